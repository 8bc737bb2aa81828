"""Host-side logic of the module mirrors, exercised on CPU through the test stand-in kernels (tests/ops_cpu.py).

What this pins (no GPU needed): constructor signatures and state-dict keys, the hand-written backward orchestration in
gzsl_text_amd/engine.py (what is saved, how gradients are routed through label attention, the GRU/GCN label
encoders, conv and embedding) against the golden vectors captured from the reference, and that the product refuses to
compute without its HIP library / on CPU tensors.
"""
import numpy as np
import pytest
import torch

from gzsl_text_amd.modules.icd_modules import ConvLabelAttnGAN, ConvLabelAttnModel
from tests import helpers
from tests.conftest import load_golden
from tests.golden import synth
from tests.ops_cpu import CpuOps


@pytest.mark.parametrize('name', list(synth.BASE_CASES))
@pytest.mark.parametrize('train', [False, True])
def test_engine_orchestration_matches_reference(name, train):
    g = load_golden(f"{name}_{'train' if train else 'eval'}")
    case = synth.synth_base(**synth.BASE_CASES[name])
    model = helpers.build_model(case, ops=CpuOps())
    helpers.check_base_against_golden(model, case, g, train)


def test_state_dict_keys_match_reference_layout():
    case = synth.synth_base(**synth.BASE_CASES['base_gate_tiny'])
    model = helpers.build_model(case, ops=CpuOps())
    keys = set(model.state_dict_to_save().keys())
    expect = {'class_margin', 'emb.embed.weight', 'graph_label_encoder.gru.weight_ih',
              'graph_label_encoder.gru.weight_hh', 'graph_label_encoder.gru.bias_ih',
              'graph_label_encoder.gru.bias_hh', 'output_proj.weight', 'output_proj.bias', 'conv_modules.0.weight',
              'conv_modules.0.bias', 'proj.weight', 'proj.bias'}  # SURVEY.md 3.4 (observed on the reference)
    assert keys == expect
    assert model.name == 'convattn_gategnn_cm2.model'
    d = case['dims']
    assert model.state_dict()['conv_modules.0.weight'].shape == (d['F'], d['E'], d['k'])
    assert model.feat_size == d['E'] + d['H'] and model.embed_size == d['E']


def test_product_has_no_cpu_path():
    case = synth.synth_base(**synth.BASE_CASES['base_gate_tiny'])
    model = helpers.build_model(case)  # default provider = HIP
    x, y, mask = helpers.batch_tensors(case)
    with pytest.raises(RuntimeError):
        model(x, y, mask)


def test_gan_variant_matches_reference():
    g = load_golden('gan_model_tiny')
    case = synth.synth_base(**synth.BASE_CASES['base_gate_tiny'])
    model = helpers.build_model(case, cls=ConvLabelAttnGAN, ops=CpuOps())
    model.load_pretrained_state_dict({k: torch.from_numpy(v.copy()) for k, v in case['params'].items()})
    assert [n for n, p in model.named_parameters() if p.requires_grad] == ['output_fc.weight']
    assert model.name == 'gan_convattn_gategnn_cm2.model' and model.pretrain_name == 'convattn_gategnn_cm2.model'
    model.eval()
    avg, graph = model.get_froze_label_emb()
    helpers.close(avg, g['frozen_avg'], 1e-5, 1e-6, 'frozen avg')
    helpers.close(graph, g['frozen_graph'], 1e-5, 2e-6, 'frozen graph')
    model.init_output_fc(torch.from_numpy(g['fc0']))
    x, y, mask = helpers.batch_tensors(case)
    model.zero_grad()
    logits, loss = model(x, y, mask)
    loss.backward()
    helpers.close(logits, g['full_logits'], 1e-5, 2e-5, 'full logits')
    helpers.close(loss, g['full_loss'], 1e-5, 1e-5, 'full loss')
    helpers.close(model.output_fc.weight.grad, g['full_dfc'], 1e-4, 1e-6, 'full dfc')
    model.zero_grad()
    logits, loss = model(x, y, mask, label_indices=torch.from_numpy(g['sub']))
    loss.backward()
    helpers.close(logits, g['sub_logits'], 1e-5, 2e-5, 'subset logits')
    helpers.close(loss, g['sub_loss'], 1e-5, 1e-5, 'subset loss')
    helpers.close(model.output_fc.weight.grad, g['sub_dfc'], 1e-4, 1e-6, 'subset dfc')
    with torch.no_grad():
        direct, attn = model.forward_real_feats(x, mask)
    helpers.close(direct, g['real_direct'], 1e-5, 2e-5, 'forward_real_feats direct')
    helpers.close(attn, g['real_attn'], 1e-5, 2e-5, 'forward_real_feats attn')
    # forward_final + AdamW, two steps (finetune.py:283-285)
    from gzsl_text_amd.modules.common_module import AdamW
    opt = AdamW(model.output_fc.parameters(), lr=1e-3, weight_decay=5e-4, betas=(0.5, 0.999), ops=CpuOps())
    feats, labs = torch.from_numpy(g['ff_feats']), torch.from_numpy(g['ff_labels'])
    for step in (1, 2):
        model.zero_grad()
        loss = model.forward_final(feats, labs) / feats.shape[0]
        loss.backward()
        helpers.close(loss, g[f'ff_loss{step}'], 1e-5, 1e-5, f'forward_final loss {step}')
        helpers.close(model.output_fc.weight.grad, g[f'ff_dfc{step}'], 1e-4, 1e-6, f'forward_final grad {step}')
        opt.step()
        helpers.close(model.output_fc.weight, g[f'ff_fc_after{step}'], 1e-5, 1e-6, f'AdamW params after {step}')


@pytest.mark.parametrize('name', list(synth.GAN_TRAIN_CASES))
def test_gan_variant_train_mode_matches_reference(name):
    helpers.check_gan_variant_train_mode(name, load_golden('gan_model_train_' + name), ops=CpuOps())


def test_pointer_helper_rejects_tensor_on_another_device(monkeypatch):
    """launches go to the CURRENT device's stream, so a tensor living on another device must raise instead
    of being dereferenced by the wrong GPU.  No GPU here: stand-in tensor objects + a patched current device."""
    from types import SimpleNamespace
    from gzsl_text_amd import hip

    def fake(index, contiguous=True):
        return SimpleNamespace(is_cuda=True, device=SimpleNamespace(index=index), is_contiguous=lambda: contiguous,
                               data_ptr=lambda: 0x1000 + index)

    monkeypatch.setattr(hip, '_current_device', lambda: 0)
    assert hip.ptr(fake(0)) == 0x1000 and hip.dptr(fake(0, contiguous=False)) == 0x1000 and hip.ptr(None) is None
    with pytest.raises(RuntimeError, match='current device'):
        hip.ptr(fake(1))
    with pytest.raises(RuntimeError, match='current device'):
        hip.dptr(fake(1, contiguous=False))
    with pytest.raises(RuntimeError, match='contiguous'):
        hip.ptr(fake(0, contiguous=False))
    monkeypatch.setattr(hip, '_current_device', lambda: 1)
    assert hip.ptr(fake(1)) == 0x1001
    with pytest.raises(RuntimeError, match='CPU tensor'):
        hip.ptr(torch.zeros(3))
