"""BASELINE.json configs[0] shapes end to end (GPU test and, through the CPU stand-in, orchestration test): a synthetic `$ICD_DATA_DIR` (tools/make_synth_icd_dir.py) ->
model built as train_base.py:76-90 builds it (gate encoder, LDAM C=2, Adam) -> one training epoch through the device
batch iterator -> dev scoring through the device counters.  The drivers themselves are not part of the build (and the
reference's were not run: DESIGN.md section 2), so the check is against the oracle evaluated at the TRAINED parameters:
dev logits / loss / thresholded metrics must agree, which covers training-updated weights of every parameter group."""
import numpy as np
import torch

from oracle import base_model as om
from oracle import data_dir as od
from oracle import eval_path as oe
from tests.helpers import close
from tools.make_synth_icd_dir import generate

T = 200


def run(tmp_path, DEV, ops=None):
    from gzsl_text_amd.modules.common_module import get_loss_fn, get_optimizer, get_scheduler
    from gzsl_text_amd.modules.icd_modules import ConvLabelAttnModel
    from gzsl_text_amd.utils.helper import iterate_minibatch
    from gzsl_text_amd.utils.metrics import EvalCounters

    root = str(tmp_path)
    generate(root, T=T)
    d = od.load_everything(root, T)
    L = len(d['targets'])
    adj = torch.from_numpy(d['adj'])
    model = ConvLabelAttnModel(d['word_emb'], torch.from_numpy(d['code_idx_matrix']), torch.from_numpy(d['code_idx_mask']),
                               adj.to_sparse().to(DEV), torch.from_numpy(d['num_neighbors']).to(DEV),
                               get_loss_fn('bce', reduction='sum'), graph_encoder='gate', eval_code_size=L,
                               target_count=d['target_count'], C=2.0)
    model.to(DEV)
    if ops is not None:
        model.use_ops(ops)
    optimizer = get_optimizer(1e-3, model, weight_decay=1e-5)
    scheduler = get_scheduler(optimizer, 1, ratios=(0.6, 0.85))
    before = {k: v.detach().clone() for k, v in model.named_parameters()}

    train_x, train_y, train_masks = d['train']
    np.random.seed(0)
    model.train()
    losses = []
    for x, y, mask, _ in iterate_minibatch(train_x, list(train_y), train_masks, L, batch_size=8, shuffle=True, device=DEV, ops=ops):
        optimizer.zero_grad()
        logits, loss = model.forward(x, y, mask)
        loss.mean().backward()
        optimizer.step()
        losses.append(float(loss.detach()))
    scheduler.step()
    assert len(losses) == 8 and np.isfinite(losses).all()
    for k, v in model.named_parameters():
        assert (not v.requires_grad) or not torch.equal(v, before[k]), f'{k} was not updated'

    dev_x, dev_y, dev_masks = d['dev']
    counters = EvalCounters(L, device=DEV, ops=ops)
    got_logits, got_losses, got_y = [], [], []
    model.eval()
    with torch.no_grad():
        for x, y, mask, _ in iterate_minibatch(dev_x, list(dev_y), dev_masks, L, batch_size=8, shuffle=False, device=DEV, ops=ops):
            logits, loss = model.forward(x, y, mask)
            counters.update(logits, y)
            got_logits.append(logits.cpu().numpy()), got_losses.append(float(loss.detach())), got_y.append(y.cpu().numpy())

    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict_to_save().items()}
    rows, cols = np.nonzero(d['adj'])
    bufs = {'adj_matrix': (rows.astype(np.int64), cols.astype(np.int64), d['adj'][rows, cols], d['adj'].shape[0]),
            'code_idx_matrix': d['code_idx_matrix'], 'code_idx_mask': d['code_idx_mask'], 'num_neighbors': d['num_neighbors']}
    margin = om.class_margin_from_count(d['target_count'])
    y_dense = oe.multihot(list(dev_y), L)
    assert np.array_equal(np.concatenate(got_y), y_dense)
    want_logits = []
    for i in range(0, len(dev_x), 8):
        ref = om.base_forward_backward(params, bufs, dev_x[i:i + 8], y_dense[i:i + 8], dev_masks[i:i + 8], L, margin, 2.0,
                                       graph_encoder='gate', drop_seed=None, need_grad=False)
        want_logits.append(ref['logits'])
        close(got_losses[i // 8], ref['loss'], 1e-5, 1e-4, f'dev loss, batch {i // 8}')
    want_logits = np.concatenate(want_logits)
    close(np.concatenate(got_logits), want_logits, 1e-5, 2e-5, 'dev logits at the trained parameters')
    score = oe.sigmoid(want_logits)
    assert np.array_equal(counters.host_counts(), oe.confusion_counts(score, y_dense))
    for name, idx in (('all', None), ('few', d['indices']['few']), ('zero', d['indices']['zero'])):
        got = counters.metrics(idx)
        cols_ = slice(None) if idx is None else idx
        want = oe.thresholded_metrics(score[:, cols_], y_dense[:, cols_])
        for k in want:
            assert np.isclose(got[k], want[k], rtol=1e-9, atol=0, equal_nan=True), (name, k, got[k], want[k])
