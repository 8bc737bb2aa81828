"""tools/make_synth_icd_dir.py writes a `$ICD_DATA_DIR` the reference's loaders can parse.  The reader is
the restatement in oracle/data_dir.py (parity unpinned: no reference fixture exists for these files and the drivers
were not run); this test checks every constraint the reference's readers impose on the files."""
import os
import pickle

import numpy as np
import pytest

from oracle import data_dir as od
from tools.make_synth_icd_dir import generate

T = 200


@pytest.fixture(scope='module')
def root(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('icd'))
    generate(out, T=T)
    return out


def test_every_file_constant_py_asserts_exists(root):
    for rel in ('resources/vocab_to_ix.pkl', 'resources/icd_word_emb.pkl', 'resources/mimic3_note_keywords.pkl',
                'resources/icd_code_hierarchy.txt', 'resources/code_desc_vocab.npz'):     # constant.py:46-48
        assert os.path.isfile(os.path.join(root, rel)), rel
    assert os.path.isdir(os.path.join(root, 'resources/splits'))                          # constant.py:50-52
    with open(os.path.join(root, 'resources/vocab_to_ix.pkl'), 'rb') as f:
        vocab = pickle.load(f)
    assert '<NUM>' in vocab and '<UNK>' in vocab


def test_loaders_restatement_accepts_the_directory(root):
    d = od.load_everything(root, T)
    L, N = len(d['targets']), len(d['ext_targets'])
    assert L == 50 and N > L and d['missed'] > 0
    assert d['adj'].shape == (N, N) and np.array_equal(d['adj'], d['adj'].T) and d['adj'].diagonal().sum() == 0
    assert (d['num_neighbors'][:L] > 0).sum() == L - d['missed']       # orphans have no neighbours, the rest do
    V1, E = d['word_emb'].shape
    assert E == 200 and d['word_emb'].dtype == np.float32 and not d['word_emb'][0].any()
    assert d['code_idx_matrix'].shape[0] == N and d['code_idx_matrix'].max() < V1 and d['code_idx_mask'].sum(1).min() >= 2
    assert ((d['code_idx_matrix'] != 0) == (d['code_idx_mask'] != 0)).all()
    for split, n in (('train', 64), ('dev', 32), ('test', 32)):
        x, y, mask = d[split]
        assert x.shape == (n, T) and mask.shape == (n, T) and y.shape == (n, 6)   # rectangular: no ragged np.asarray
        assert x.min() >= 0 and x.max() < V1 and ((x != 0) <= (mask != 0)).all()
        assert y.min() >= 0 and y.max() < L
    lens = d['train'][2].sum(1)
    assert (lens < T).any() and (lens == T).any()                      # both branches of preload_data
    ind = d['indices']
    assert len(ind['freq']) > 0 and set(range(40, 45)) <= set(ind['few']) and set(range(45, 50)) <= set(ind['zero'])
    assert all(d['target_count'][i] == 0 for i in ind['zero']) and all(0 < d['target_count'][i] <= 5 for i in ind['few'])


def test_parent_is_substring_and_neighbours_resolve(root):
    d = od.load_everything(root, T)
    with open(os.path.join(root, 'resources/icd_code_hierarchy.txt')) as f:
        for line in f:
            code, rest = line[:-1].split(':')
            nbs = rest.split(' ')
            assert nbs != [''] and all(nb in d['ext_targets'] for nb in nbs)      # data_loader.py:63 would KeyError
            assert sum(nb in code for nb in nbs) <= 1                             # at most one parent (:64)


def test_keyword_and_desc_vocab_files(root):
    d = od.load_everything(root, T)
    with np.load(os.path.join(root, 'resources/code_desc_vocab.npz')) as f:
        words, emb = f['arr_0'], f['arr_1']
    assert words[0] == 0 and np.array_equal(words, np.sort(np.unique(d['code_idx_matrix'])))   # data_loader.py:121,128
    assert np.array_equal(emb, d['word_emb'][words])
    with open(os.path.join(root, 'resources/mimic3_note_keywords.pkl'), 'rb') as f:
        kw = pickle.load(f)
    names, paths, labels = od.read_split(root, 'train')
    for path, codes in zip(paths, labels):
        per_code = kw[os.path.split(path)[-1]]                                     # keywords_loader.py:30
        assert set(per_code) == {int(d['targets'][c]) for c in codes}
        ids, scores = next(iter(per_code.values()))[:2]                            # feature_loader.py:93
        assert ids.max() < len(words) and (np.diff(scores) <= 0).all() and (scores >= 0.1).any()


def test_cfg1_flow_orchestration_on_cpu_stand_in(tmp_path):
    """The same flow as tests/test_gpu_cfg1_end_to_end.py with tests/ops_cpu.py injected: host logic only."""
    from tests import cfg1_flow
    from tests.ops_cpu import CpuOps
    cfg1_flow.run(tmp_path, 'cpu', CpuOps())
