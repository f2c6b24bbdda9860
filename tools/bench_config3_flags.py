"""Development aid: the headline loop of bench.py (config 3, no extras) with an experimental build of the library (extra hipcc flags).
usage (GPU box): python tools/bench_config3_flags.py [-DFLAG ...]"""
import json, os, subprocess, sys
sys.path.insert(0, '.')
flags = sys.argv[1:]
env = dict(os.environ)
if flags:
    from chinesecheckersagent_amd import build as B
    so = os.path.join('chinesecheckersagent_amd', 'libccsp_exp.so')
    subprocess.check_call(['hipcc'] + B.FLAGS + ['-shared'] + flags + ['-o', so] + [os.path.join(B.CSRC, f) for f in B.SOURCES])
    env['CCSP_LIB'] = os.path.abspath(so)
for _ in range(2):
    out = subprocess.run([sys.executable, 'bench.py', '--no-extras', '--steps', '48', '--warmup', '5'], env=env, capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if not lines:
        print(out.stderr[-1500:]); sys.exit(1)
    d = json.loads(lines[0])
    print(flags, 'value %.4g  ms/sim-step %.5f  in-pipeline ms/launch %.5f  alone %.5f' %
          (d['value'], d['ms_per_sim_step'], d['roofline']['in_pipeline_ms_per_launch'], d['roofline']['avg_launch_ms']))
