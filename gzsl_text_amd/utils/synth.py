"""Seeded synthetic MIMIC-shaped inputs and parameters (pure numpy; no reference code).

Shared by tests/golden/make_golden.py (which feeds them to the reference to capture golden
outputs), by the parity tests, by bench.py and by __graft_entry__.smoke(); `build_model` /
`batch_tensors` turn a case into the module mirror exactly as train_base.py:76-84 builds it.  Uses the legacy
`np.random.RandomState` streams, which NumPy freezes across versions, so regenerated inputs
match the ones the fixtures were captured on; every fixture also stores an input checksum.

Tensor formats follow the drivers' input contract:
  x int64 [B,T] (0 = pad), mask fp32 [B,T], y fp32 multi-hot [B,L]   (utils/helper.py:106-132)
  code_idx_matrix int64 [N,Td], code_idx_mask fp32 [N,Td]            (loaders/emb_loader.py:37-60)
  adj_matrix fp32 [N,N] symmetric forest, num_neighbors = adj.sum(1)  (loaders/data_loader.py:47-72)
"""
import numpy as np


def _uniform(rs, shape, a):
    return rs.uniform(-a, a, size=shape).astype(np.float32)


def forest_adjacency(rs, N):
    """Each node (after the first) gets one parent among earlier nodes; symmetric 0/1 matrix as COO."""
    child = np.arange(1, N, dtype=np.int64)
    parent = (rs.random_sample(N - 1) * child).astype(np.int64)
    rows = np.concatenate([child, parent])
    cols = np.concatenate([parent, child])
    vals = np.ones(rows.shape[0], np.float32)
    deg = np.bincount(rows, minlength=N).astype(np.float32)
    return rows, cols, vals, deg


def coo_to_dense(rows, cols, vals, N):
    a = np.zeros((N, N), np.float32)
    a[rows, cols] = vals
    return a


def synth_base(seed, B, T, E, H, L, N, Td, V, F, k=10, encoder='gate', labels_per_note=3,
               emb_scale=0.5, zipf=False):
    rs = np.random.RandomState(seed)
    D = E + H
    p = {}
    W = (rs.standard_normal((V, E)) * emb_scale).astype(np.float32)
    p['emb.embed.weight'] = W
    p['conv_modules.0.weight'] = _uniform(rs, (F, E, k), np.sqrt(6.0 / (E * k + F * k)))
    p['conv_modules.0.bias'] = _uniform(rs, (F,), 0.1)
    p['proj.weight'] = _uniform(rs, (E, F), np.sqrt(6.0 / (E + F)))
    p['proj.bias'] = _uniform(rs, (E,), 0.1)
    p['output_proj.weight'] = _uniform(rs, (D, F), np.sqrt(6.0 / (D + F)))
    p['output_proj.bias'] = _uniform(rs, (D,), 0.1)
    if encoder == 'gate':
        assert E == H, "GRUCell(E,H) is called with a hidden state of width E (graph_module.py:106,122)"
        a = 1.0 / np.sqrt(H)
        p['graph_label_encoder.gru.weight_ih'] = _uniform(rs, (3 * H, E), a)
        p['graph_label_encoder.gru.weight_hh'] = _uniform(rs, (3 * H, H), a)
        p['graph_label_encoder.gru.bias_ih'] = _uniform(rs, (3 * H,), a)
        p['graph_label_encoder.gru.bias_hh'] = _uniform(rs, (3 * H,), a)
    elif encoder == 'conv':
        p['graph_label_encoder.gc1.fc.weight'] = _uniform(rs, (H, 2 * E), 1.0 / np.sqrt(2 * E))
        p['graph_label_encoder.gc1.fc.bias'] = _uniform(rs, (H,), 0.1)
        p['graph_label_encoder.gc2.fc.weight'] = _uniform(rs, (H, 2 * H), 1.0 / np.sqrt(2 * H))
        p['graph_label_encoder.gc2.fc.bias'] = _uniform(rs, (H,), 0.1)
    # label descriptions
    dlen = rs.randint(2, Td + 1, size=N)
    idx = rs.randint(1, V, size=(N, Td)).astype(np.int64)
    cmask = (np.arange(Td)[None, :] < dlen[:, None]).astype(np.float32)
    idx = idx * cmask.astype(np.int64)
    rows, cols, vals, deg = forest_adjacency(rs, N)
    # notes
    lens = rs.randint(max(k + 1, T // 2), T + 1, size=B)
    lens[0] = T
    if zipf:
        ids = np.minimum(rs.zipf(1.1, size=(B, T)), V - 1).astype(np.int64)
    else:
        ids = rs.randint(1, V, size=(B, T)).astype(np.int64)
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)
    x = ids * mask.astype(np.int64)
    y = np.zeros((B, L), np.float32)
    for b in range(B):
        y[b, rs.choice(L, size=min(labels_per_note, L), replace=False)] = 1.0
    target_count = np.floor(rs.pareto(1.0, size=L) * 3).astype(np.float32)  # long tail, some zeros
    bufs = dict(code_idx_matrix=idx, code_idx_mask=cmask, adj_matrix=(rows, cols, vals, N),
                num_neighbors=deg)
    return dict(params=p, bufs=bufs, x=x, y=y, mask=mask, target_count=target_count,
                dims=dict(B=B, T=T, E=E, H=H, L=L, N=N, Td=Td, V=V, F=F, k=k, D=D), encoder=encoder)


def synth_gan(seed, Bg, D, Ls, ngh, ndh, E, Vk, top_k, n_iters, add_zero=True):
    """Inputs for one WGAN-GP step: n_iters critic iterations + 1 generator iteration."""
    rs = np.random.RandomState(seed)
    g = {'fc1.weight': _uniform(rs, (ngh, 2 * Ls), 1 / np.sqrt(2 * Ls)), 'fc1.bias': _uniform(rs, (ngh,), 0.05),
         'fc2.weight': _uniform(rs, (D, ngh), 1 / np.sqrt(ngh)), 'fc2.bias': _uniform(rs, (D,), 0.05)}
    d = {'fc1.weight': _uniform(rs, (ndh, D + Ls), 1 / np.sqrt(D + Ls)), 'fc1.bias': _uniform(rs, (ndh,), 0.05),
         'fc2.weight': _uniform(rs, (1, ndh), 1 / np.sqrt(ndh)), 'fc2.bias': _uniform(rs, (1,), 0.05)}
    kw = {'proj_fc.weight': _uniform(rs, (E, D), 1 / np.sqrt(D)), 'proj_fc.bias': _uniform(rs, (E,), 0.05),
          'word_emb': (rs.standard_normal((Vk, E)) * 0.5).astype(np.float32)}
    iters = []
    for _ in range(n_iters + 1):
        real = np.maximum(rs.standard_normal((Bg, D)), 0).astype(np.float32)
        lab = (rs.standard_normal((Bg, Ls)) * 0.5).astype(np.float32)
        noise = rs.standard_normal((Bg, Ls)).astype(np.float32)
        alpha = rs.random_sample((Bg, 1)).astype(np.float32)
        sim = np.ones(Bg, np.float32)
        if add_zero:
            z = rs.random_sample(Bg) < 0.25
            sim = np.where(z, rs.uniform(0.5, 1.0, Bg), 1.0).astype(np.float32)
        iters.append(dict(real=real, lab=lab, noise=noise, alpha=alpha, sim=sim))
    kw_idx = np.stack([rs.choice(Vk, size=top_k, replace=False) for _ in range(Bg)]).astype(np.int64)
    kw_mask = rs.uniform(0.1, 1.0, (Bg, top_k)).astype(np.float32)
    kw_mask[:, top_k - 2:] *= (rs.random_sample((Bg, 2)) < 0.5)  # some padded keyword slots
    return dict(g=g, d=d, kw=kw, iters=iters, kw_idx=kw_idx, kw_mask=kw_mask,
                dims=dict(Bg=Bg, D=D, Ls=Ls, ngh=ngh, ndh=ndh, E=E, Vk=Vk, top_k=top_k))


def checksum(*arrays):
    """Order-sensitive float64 checksum used to detect drift of regenerated inputs."""
    s = 0.0
    for i, a in enumerate(arrays):
        a = np.asarray(a, np.float64).reshape(-1)
        s += float((a * (1.0 + (np.arange(a.size) % 97) / 97.0)).sum()) * (i + 1)
    return np.float64(s)


BASE_CASES = {
    # name: kwargs for synth_base (+ train seed where dropout is on)
    'base_gate_tiny': dict(seed=0, B=3, T=40, E=8, H=8, L=12, N=20, Td=6, V=50, F=24),
    'base_conv_tiny': dict(seed=1, B=2, T=33, E=8, H=12, L=9, N=14, Td=5, V=40, F=20, encoder='conv'),
    'base_gate_medium': dict(seed=2, B=2, T=300, E=100, H=100, L=256, N=300, Td=12, V=500, F=200,
                             labels_per_note=8),
    # E = H = 200 (what the drivers build from the shipped resources): the E > 100 kernel variants, 3 token tiles of 64
    'base_gate_e200': dict(seed=3, B=2, T=150, E=200, H=200, L=70, N=90, Td=8, V=120, F=64, labels_per_note=5),
    # 8 notes, 19 labels: the 8-rank sharding tests (batch-split encoder, label ranges of 3 and 2 rows)
    'base_gate_w8': dict(seed=4, B=8, T=36, E=8, H=8, L=19, N=26, Td=5, V=60, F=24),
}
DROP_SEED = 1234

# ConvLabelAttnGAN.forward in TRAIN mode (finetune.py:258,271: both dropouts live): base-case kwargs + the size of the
# `label_indices` subset.  'e200' is the width the drivers build (E = H = 200, D = 400) at T' = 291.
GAN_TRAIN_CASES = {
    'tiny': dict(base=dict(seed=0, B=3, T=40, E=8, H=8, L=12, N=20, Td=6, V=50, F=24), n_sub=5),
    'e200': dict(base=dict(seed=5, B=2, T=300, E=200, H=200, L=256, N=300, Td=8, V=400, F=200, labels_per_note=8),
                 n_sub=37),
}


def gan_train_inputs(name):
    """(case, fc0 [N,D], sub int64 [n_sub]) of a GAN_TRAIN_CASES entry: the classifier table and the label subset are
    seeded here (not derived from reference outputs), so tests regenerate them."""
    spec = GAN_TRAIN_CASES[name]
    case = synth_base(**spec['base'])
    d = case['dims']
    rs = np.random.RandomState(1000 + spec['base']['seed'])
    fc0 = (rs.standard_normal((d['N'], d['D'])) * 0.15).astype(np.float32)
    sub = rs.choice(d['L'], size=spec['n_sub'], replace=False).astype(np.int64)  # unsorted, as finetune.py's syn_indices
    return case, fc0, sub


def build_model(case, cls=None, device='cpu', ops=None, sparse=True, C=2.0):
    """Construct the module mirror exactly as train_base.py:76-84 does and load the case's parameters."""
    import torch
    from ..modules.common_module import get_loss_fn
    from ..modules.icd_modules import ConvLabelAttnModel
    cls = ConvLabelAttnModel if cls is None else cls
    d = case['dims']
    bufs = case['bufs']
    rows, cols, vals, N = bufs['adj_matrix']
    adj = torch.from_numpy(coo_to_dense(rows, cols, vals, N))
    model = cls(case['params']['emb.embed.weight'].copy(), torch.from_numpy(bufs['code_idx_matrix']),
                torch.from_numpy(bufs['code_idx_mask']), adj, torch.from_numpy(bufs['num_neighbors']),
                get_loss_fn('bce', reduction='sum'), num_filters=d['F'], kernel_sizes=(d['k'],),
                eval_code_size=d['L'], label_hidden_size=d['H'], graph_encoder=case['encoder'],
                target_count=case['target_count'].copy(), C=C)
    sd = model.state_dict()
    for k, v in case['params'].items():
        assert k in sd and tuple(sd[k].shape) == v.shape, (k, v.shape)
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)
    if sparse:
        model.adj_matrix = model.adj_matrix.to_sparse()  # train_base.py:79
    model.to(device)
    if ops is not None:
        model.use_ops(ops)
    return model


def batch_tensors(case, device='cpu'):
    import torch
    return tuple(torch.from_numpy(case[k]).to(device) for k in ('x', 'y', 'mask'))
