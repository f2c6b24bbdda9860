"""The reference's offline data utilities on the pure-Python HDF5 reader/writer (no h5py needed):

    combine_train_data(board_x, pi_y, v_y, first_version, last_version, save_dir, pref)      combine_data.py:5-36
    save_combined(board_x, pi_y, v_y, path='combined.h5')                                    combine_data.py:39-44
    count_items(v_y) / get_train_label_count(path)                                           count_labels.py:5-22
    get_weights(model_path, out_path=None)        whole-model file -> weights-only file      get_model_weights.py:5-10

    python -m chinesecheckersagent_amd.datatools combine <dir> <pref> <first> <last>         -> combined.h5
    python -m chinesecheckersagent_amd.datatools count <file.h5>
    python -m chinesecheckersagent_amd.datatools weights <model.h5> [<out.h5>]
"""
import os
import sys

import numpy as np

from .h5lite import H5File, write_datasets


def _read(path):
    f = H5File(path)
    return np.array(f.get('board_x')), np.array(f.get('pi_y')), np.array(f.get('v_y'))


def combine_train_data(board_x, pi_y, v_y, first_version, last_version, save_dir, pref):
    """the given samples (if all three are non-empty) plus the files '{save_dir}/{pref}{i}.h5', i = first..last (i >= 0), that
    exist -> (board_x, pi_y, v_y, number of sources) as arrays, or ([], [], [], 0) when there is nothing"""
    sources = [(np.asarray(board_x), np.asarray(pi_y), np.asarray(v_y))] if min(len(board_x), len(pi_y), len(v_y)) > 0 else []
    names = ('{}/{}{}.h5'.format(save_dir, pref, i) for i in range(max(first_version, 0), last_version + 1))
    sources += [_read(name) for name in names if os.path.exists(name)]
    if not sources:
        return [], [], [], 0
    bx, py, vy = zip(*sources)
    return np.concatenate(bx), np.concatenate(py), np.concatenate(vy), len(sources)


def save_combined(board_x, pi_y, v_y, path='combined.h5'):
    write_datasets(path, [('board_x', np.asarray(board_x, dtype=np.float64)), ('pi_y', np.asarray(pi_y, dtype=np.float64)),
                          ('v_y', np.asarray(v_y, dtype=np.int64))])
    return path


def count_items(v_y):
    count = {}
    for val in np.asarray(v_y).tolist():
        count[val] = count.get(val, 0) + 1
    return count


def get_train_label_count(path):
    return count_items(_read(path)[2])


def get_weights(model_path, out_path=None):
    """get_model_weights.py: a Keras whole-model file (model.save) -> a weights-only file in the layout load_weights
    reads ('<name>-weights.h5' next to it by default); layer names come out canonical (conv2d_1.., dense_1)"""
    from .model import read_keras_weights
    from .train import keras_layer_names
    w = read_keras_weights(model_path)
    layers = [(ln, [(wn, np.asarray(w['%s/%s' % (ln, wn)])) for wn in wns]) for ln, wns in keras_layer_names()]
    if out_path is None:
        out_path = (model_path[:-3] if model_path.endswith('.h5') else model_path) + '-weights.h5'
    from .h5lite import write_keras_weights
    write_keras_weights(out_path, layers)
    return out_path


if __name__ == '__main__':
    a = sys.argv[1:]
    if len(a) == 5 and a[0] == 'combine':
        bx, py, vy, n = combine_train_data([], [], [], int(a[3]), int(a[4]), a[1], a[2])
        if n:
            print('%d files -> %s (%d samples)' % (n, save_combined(bx, py, vy), len(vy)))
        else:
            print('no data')
    elif len(a) == 2 and a[0] == 'count':
        print(get_train_label_count(a[1]))
    elif len(a) in (2, 3) and a[0] == 'weights':
        print(get_weights(*a[1:]))
    else:
        print(__doc__)
