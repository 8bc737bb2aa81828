"""Mirror of the reference's modules/graph_module.py: ICD-hierarchy label encoders (graph_module.py:32-129).

The classes hold the same parameters under the same names (`gru.*`, `gc1.fc.*`, `gc2.fc.*`) so checkpoints load
unchanged.  Inside ConvLabelAttnModel they are parameter holders: the model's fused autograd function runs the
encoder through BaseEngine.label_fwd / label_bwd.  Called directly (2-D input, the only form the drivers use) they
run the same engine code without gradient tracking.
"""
import torch
import torch.nn as nn

from ..engine import BaseEngine, GraphBuffers


def _standalone(kind, module, x, adj_matrix, num_neighbors, prefix):
    if x.dim() != 2:
        raise NotImplementedError('only the 2-D (label) branch is on the hot path (graph_module.py:61,105)')
    if getattr(module, '_ops', None) is None:
        from ..ops_hip import HipOps
        module._ops = HipOps()   # one provider per module, not per call
    eng = BaseEngine(module._ops, graph_encoder=kind, steps=getattr(module, 'steps', 2))
    P = {prefix + k: v.detach() for k, v in module.named_parameters()}
    gb = GraphBuffers.__new__(GraphBuffers)
    from ..engine import build_csr, EPS_DEG
    gb.csr, gb.csr_t = build_csr(adj_matrix)
    nn_ = num_neighbors if num_neighbors is not None else torch.ones(x.shape[0], device=x.device) - EPS_DEG
    gb.inv_deg = (1.0 / (nn_.float() + EPS_DEG)).contiguous()
    x = x.detach().contiguous()
    ops = eng.ops
    if kind == 'gate':
        h = x
        for _ in range(eng.steps):
            u = ops.spmm(gb.csr, h, post=gb.inv_deg)
            gi = ops.linear(u, P[prefix + 'gru.weight_ih'], P[prefix + 'gru.bias_ih'])
            gh = ops.linear(h, P[prefix + 'gru.weight_hh'], P[prefix + 'gru.bias_hh'])
            h = ops.gru_gate_fwd(gi, gh, h)
        return torch.cat([x, h], -1) if module.concat else h
    from ..hip import ACT_RELU
    f = x
    for name in ('gc1', 'gc2'):
        nb = ops.spmm(gb.csr, f, post=gb.inv_deg)
        f = ops.linear(torch.cat([f, nb], 1), P[f'{prefix}{name}.fc.weight'], P[f'{prefix}{name}.fc.bias'],
                       act=ACT_RELU)
    return torch.cat([f, x], -1) if module.concat else f


class ConvGraphLayer(nn.Module):
    def __init__(self, input_size, hidden_size, sparse=True, activation=torch.relu):
        super(ConvGraphLayer, self).__init__()
        self.fc = nn.Linear(input_size * 2, hidden_size)
        self.sparse = sparse
        self.activation = activation


class ConvGraphEncoder(nn.Module):
    def __init__(self, input_size, hidden_size, n_nodes, concat=True, sparse=True):
        super(ConvGraphEncoder, self).__init__()
        self.sparse = sparse
        self.N = n_nodes
        self.gc1 = ConvGraphLayer(input_size, hidden_size, sparse)
        self.gc2 = ConvGraphLayer(hidden_size, hidden_size, sparse)
        self.concat = concat
        self.output_size = hidden_size
        if concat:
            self.output_size += hidden_size

    def forward(self, x, adj_matrix, num_neighbors):
        return _standalone('conv', self, x, adj_matrix, num_neighbors, '')


class GraphGatedEncoder(nn.Module):
    def __init__(self, input_size, hidden_size, n_nodes, concat=True, steps=2):
        super(GraphGatedEncoder, self).__init__()
        self.gru = nn.GRUCell(input_size, hidden_size)
        self.steps = steps
        self.concat = concat
        self.output_size = hidden_size
        if concat:
            self.output_size += hidden_size

    def forward(self, x, adj_matrix, num_neighbors=None):
        return _standalone('gate', self, x, adj_matrix, num_neighbors, '')
