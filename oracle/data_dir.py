"""TEST INFRASTRUCTURE ONLY -- restatement of how the reference's loaders read `$ICD_DATA_DIR` and turn it into the
model's constructor arguments.  PARITY UNPINNED: the reference has no fixtures for these files and its
loaders / drivers were not run here, so this reader is pinned only by the format description
in the reference's sources, cited per function.  Used by tests/ to check tools/make_synth_icd_dir.py's output and to
feed the GPU end-to-end test; the product never imports it.
"""
import os
import pickle
from collections import Counter

import numpy as np


def read_split(root, mode, thresh=5):
    """data_loader.py:139-160,178-198: sorted note names of a split, their note paths and label code lists."""
    with open(os.path.join(root, 'resources', 'splits', f'{mode}_thresh{thresh}_hdam_ids.txt')) as f:
        names = sorted(line[:-1] for line in f)
    labels = []
    for name in names:
        with open(os.path.join(root, 'processed', 'label_files', f'{name}_labels.txt')) as f:
            codes = [line.split(',')[1].strip() for line in f]
        codes = ['7197' if c == '71970' else c for c in codes]          # :189-190
        labels.append([c for c in codes if c != 'V122'])                # :192-193
    paths = [os.path.join(root, 'processed', 'numpy_vectors', f'{name}_notes.npy') for name in names]
    return names, paths, labels


def codes_to_targets(all_codes):
    """helper.py:181: sorted codes -> 0..n-1."""
    return {c: i for i, c in enumerate(sorted(all_codes))}


def adjacency(root, targets):
    """data_loader.py:38-72: hierarchy-only codes are appended after the label codes in file order; symmetric 0/1."""
    hier = {}
    with open(os.path.join(root, 'resources', 'icd_code_hierarchy.txt')) as f:
        for line in f:
            code, neighbors = line[:-1].split(':')
            hier[code] = neighbors.split(' ')
    ext = dict(targets)
    for code in hier:
        if code not in ext:
            ext[code] = len(ext)
    adj = np.zeros((len(ext), len(ext)))
    missed = 0
    for code in ext:
        if code in hier:
            for nb in hier[code]:
                adj[ext[code], ext[nb]] = 1
                adj[ext[nb], ext[code]] = 1
        else:
            missed += 1
    return ext, adj, missed


def preload(paths, labels, targets, max_note_len, idx_offset=1):
    """data_loader.py:75-104: ids shifted by one (0 = padding), zero-padded or truncated to max_note_len."""
    x = np.zeros((len(paths), max_note_len), dtype=np.int64)
    mask = np.ones((len(paths), max_note_len), dtype=np.float32)
    for row, path in enumerate(paths):
        note = np.load(path) + idx_offset
        if len(note) < max_note_len:
            x[row, :len(note)] = note
            mask[row, len(note):] = 0
        else:
            x[row] = note[:max_note_len]
    y = np.asarray([[targets[c] for c in codes] for codes in labels])
    return x, y, mask


def embeddings_and_code_input(root, ext_targets):
    """emb_loader.py:25-60: two consecutive pickles; description ids right-padded with 0 to the longest description."""
    with open(os.path.join(root, 'resources', 'icd_word_emb.pkl'), 'rb') as f:   # written by our own generator
        word_emb = pickle.load(f)
        code_to_idx = pickle.load(f)
    width = max(len(v) for v in code_to_idx.values())
    order = sorted(ext_targets, key=ext_targets.get)
    idx = np.zeros((len(order), width), dtype=np.int64)
    mask = np.zeros((len(order), width), dtype=np.float32)
    for row, code in enumerate(order):
        words = code_to_idx[code]
        idx[row, :len(words)] = words
        mask[row, :len(words)] = 1
    return word_emb, idx, mask


def split_codes_by_count(train_labels, dev_labels, thresh=5):
    """helper.py:233-251: dev codes seen >= thresh times: zero-shot if never in train, few-shot if <= 5 train notes."""
    train_count, dev_count = Counter(c for l in train_labels for c in l), Counter(c for l in dev_labels for c in l)
    freq, few, zero = set(), set(), set()
    for code, n in dev_count.items():
        if n < thresh:
            continue
        (zero if code not in train_count else few if train_count[code] <= 5 else freq).add(code)
    return freq, few, zero, train_count


def load_everything(root, max_note_len):
    """train_base.py:38-79 up to the model constructor."""
    splits = {m: read_split(root, m) for m in ('train', 'dev', 'test')}
    all_codes = set(c for m in splits for l in splits[m][2] for c in l)
    targets = codes_to_targets(all_codes)
    ext, adj, missed = adjacency(root, targets)
    freq, few, zero, counter = split_codes_by_count(splits['train'][2], splits['dev'][2])
    target_count = np.zeros(len(targets), dtype=np.float32)          # helper.py:254-262
    for code, t in targets.items():
        target_count[t] = counter.get(code, 0)
    word_emb, code_idx, code_mask = embeddings_and_code_input(root, ext)
    out = {'targets': targets, 'ext_targets': ext, 'adj': adj.astype(np.float32), 'missed': missed,
           'num_neighbors': adj.sum(1).astype(np.float32), 'target_count': target_count, 'word_emb': word_emb,
           'code_idx_matrix': code_idx, 'code_idx_mask': code_mask,
           'indices': {k: sorted({targets[c] for c in v}) for k, v in (('freq', freq), ('few', few), ('zero', zero))}}
    for m in splits:
        out[m] = preload(splits[m][1], splits[m][2], targets, max_note_len)
    return out
