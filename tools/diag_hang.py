"""Diagnostic: run a tools/ script under faulthandler so that a stall shows WHERE the host is waiting (dump after N seconds, then exit).
usage: python3 tools/diag_hang.py <seconds> tools/bench_free.py [args...]"""
import faulthandler, sys
secs = float(sys.argv[1])
faulthandler.dump_traceback_later(secs, repeat=False, file=sys.stderr, exit=True)
script = sys.argv[2]
sys.argv = sys.argv[2:]
exec(compile(open(script).read(), script, 'exec'), {'__name__': '__main__', '__file__': script})
