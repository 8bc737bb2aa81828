"""Synthetic `$ICD_DATA_DIR` in the reference's on-disk formats: what petuum/gzsl_text's drivers read,
without MIMIC-III access (BASELINE.json configs[0] shapes by default).

    python tools/make_synth_icd_dir.py --out /some/dir [--n-train 64 --n-dev 32 --n-test 32 --T 200 --L 50 ...]

Files written (reader in the reference):
  resources/vocab_to_ix.pkl            dict word -> id, with <NUM>, <UNK>                   (preprocess.py:154-166)
  resources/icd_word_emb.pkl           pickle 1: float32 [V+1, E] (row 0 = padding), pickle 2: dict code -> list[int]
                                       for every code in labels U hierarchy                 (emb_loader.py:25-31,45-47)
  resources/mimic3_note_keywords.pkl   dict '<name>_notes.npy' -> dict target -> (keyword ids, scores)
                                                                        (keywords_loader.py:20-30, feature_loader.py:86-96)
  resources/icd_code_hierarchy.txt     'code:n1 n2 ...'; a neighbour that is a substring of the code is its parent
                                                                                            (data_loader.py:38-44,64)
  resources/code_desc_vocab.npz        arr_0 sorted unique description word ids incl. 0, arr_1 their embeddings
                                                                                            (data_loader.py:117-129)
  resources/splits/{train,dev,test}_thresh5_hdam_ids.txt   note names                      (data_loader.py:141)
  processed/label_files/<name>_labels.txt   'seq, code' lines                               (data_loader.py:183-194)
  processed/numpy_vectors/<name>_notes.npy  unshifted int word ids                          (data_loader.py:90)

Shape of the data: ICD-like codes '<3 digits><1 digit>' whose parent is the 3-digit prefix; frequent codes drawn from a
Zipf law, a few few-shot codes (<= 5 train notes, >= 5 dev/test notes) and zero-shot codes (never in train), a few
label codes without hierarchy entry; every note carries the same number of labels (numpy >= 1.24 refuses the ragged
`np.asarray(y)` of data_loader.py:103 otherwise); note lengths straddle --T so both the padding and the truncation
branch of preload_data are exercised.
"""
import argparse
import os
import pickle

import numpy as np


def generate(out, n_train=64, n_dev=32, n_test=32, T=200, L=50, labels_per_note=6, V=500, E=200, Td=8, seed=1234):
    rs = np.random.RandomState(seed)
    K = labels_per_note
    assert L >= 20 and K >= 3
    for d in ('resources/splits', 'processed/label_files', 'processed/numpy_vectors'):
        os.makedirs(os.path.join(out, d), exist_ok=True)

    # ---- codes and hierarchy ------------------------------------------------------------------------------------
    parents = [f'{100 + i:03d}' for i in range((L + 3) // 4)]
    leaves = [f'{parents[i // 4]}{i % 4}' for i in range(L)]
    orphan = set(leaves[3::11])                       # label codes with no hierarchy line ("missed", data_loader.py:69)
    extra_leaves = [f'{p}9' for p in parents[::3]]    # hierarchy-only nodes (never a label)
    neigh = {}

    def link(c, n):
        neigh.setdefault(c, []).append(n)
        neigh.setdefault(n, []).append(c)

    for leaf in leaves + extra_leaves:
        if leaf not in orphan:
            link(leaf, leaf[:3])
    for p in parents:                                 # grandparents: 2-char prefixes, substrings of the parent code
        if p in neigh:
            link(p, p[:2])
    with open(os.path.join(out, 'resources/icd_code_hierarchy.txt'), 'w') as f:
        for code in sorted(neigh):
            f.write(f'{code}:{" ".join(neigh[code])}\n')

    # ---- vocabulary, embeddings, code descriptions -----------------------------------------------------------------
    words = ['<NUM>', '<UNK>'] + [f'w{i}' for i in range(V - 2)]
    with open(os.path.join(out, 'resources/vocab_to_ix.pkl'), 'wb') as f:
        pickle.dump({w: i for i, w in enumerate(words)}, f)
    word_emb = (rs.standard_normal((V + 1, E)) * 0.1).astype(np.float32)
    word_emb[0] = 0
    all_codes = sorted(set(leaves) | set(neigh))
    code_to_idx = {c: [int(w) for w in rs.randint(1, V + 1, size=rs.randint(2, Td + 1))] for c in all_codes}
    code_to_idx[all_codes[0]] = [int(w) for w in rs.randint(1, V + 1, size=Td)]   # one description of full length
    with open(os.path.join(out, 'resources/icd_word_emb.pkl'), 'wb') as f:
        pickle.dump(word_emb, f)
        pickle.dump(code_to_idx, f)
    desc_words = np.unique(np.concatenate([[0]] + [np.asarray(v) for v in code_to_idx.values()])).astype(np.int64)
    np.savez(os.path.join(out, 'resources/code_desc_vocab.npz'), desc_words, word_emb[desc_words])

    # ---- notes, labels, splits, keywords -------------------------------------------------------------------------
    n_freq = L - 10
    few, zero = leaves[n_freq:n_freq + 5], leaves[n_freq + 5:]
    zipf = 1.0 / np.arange(1, n_freq + 1) ** 1.1
    zipf /= zipf.sum()
    targets = {c: i for i, c in enumerate(sorted(leaves))}   # what codes_to_index_labels assigns (helper.py:181)
    keywords = {}
    seq = 0
    for split, n in (('train', n_train), ('dev', n_dev), ('test', n_test)):
        names = []
        for i in range(n):
            name = f'{split}{i:04d}'
            names.append(name)
            forced = []
            if split == 'train':
                if i < 15:
                    forced = [few[i % 5]]                  # 3 train notes per few-shot code
            else:
                forced = [few[i % 5], zero[(i // 2) % 5]]  # >= 5 dev/test notes each at the default sizes
            pool = rs.choice(n_freq, size=K - len(forced), replace=False, p=zipf)
            codes = forced + [leaves[j] for j in pool]
            with open(os.path.join(out, f'processed/label_files/{name}_labels.txt'), 'w') as f:
                for c in codes:
                    seq += 1
                    f.write(f'{seq}, {c}\n')
            length = int(np.clip(rs.lognormal(np.log(T * 0.7), 0.5), 12, T * 2))
            note = (rs.zipf(1.3, size=length) % V).astype(np.int64)
            np.save(os.path.join(out, f'processed/numpy_vectors/{name}_notes.npy'), note)
            keywords[f'{name}_notes.npy'] = {
                int(targets[c]): (rs.randint(1, len(desc_words), size=20).astype(np.int64),
                                  np.sort(rs.uniform(0.05, 1.0, size=20).astype(np.float32))[::-1].copy())
                for c in codes}
        with open(os.path.join(out, f'resources/splits/{split}_thresh5_hdam_ids.txt'), 'w') as f:
            for name in names:
                f.write(name + '\n')
    with open(os.path.join(out, 'resources/mimic3_note_keywords.pkl'), 'wb') as f:
        pickle.dump(keywords, f)
    return {'label_codes': len(leaves), 'orphans': len(orphan), 'graph_nodes': len(all_codes), 'V': V, 'E': E}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--n-train', type=int, default=64)
    ap.add_argument('--n-dev', type=int, default=32)
    ap.add_argument('--n-test', type=int, default=32)
    ap.add_argument('--T', type=int, default=200, help='the --max_note_len the drivers will be run with')
    ap.add_argument('--L', type=int, default=50, help='label codes (last 10: 5 few-shot, 5 zero-shot)')
    ap.add_argument('--labels-per-note', type=int, default=6)
    ap.add_argument('--V', type=int, default=500)
    ap.add_argument('--E', type=int, default=200, help='embedding size (gate encoder needs E == label_hidden_size == 200)')
    ap.add_argument('--Td', type=int, default=8, help='longest code description')
    ap.add_argument('--seed', type=int, default=1234)
    a = ap.parse_args()
    info = generate(a.out, a.n_train, a.n_dev, a.n_test, a.T, a.L, a.labels_per_note, a.V, a.E, a.Td, a.seed)
    print(f'wrote {a.out}: {info}')


if __name__ == '__main__':
    main()
