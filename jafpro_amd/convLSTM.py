"""ConvLSTM on the fused HIP cell kernel.  Mirrors the call surface of the reference's
src/convLSTM.py (ConvLSTMCell :7-63, ConvLSTM :66-165): same constructor arguments, the same
``forward(input, hidden_state=None) -> (layer_output, last_state_list)`` contract and the same
state_dict keys (``cell_list.{i}.conv.weight|bias``).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


class _GateConv(nn.Module):
    """Parameter holder named like nn.Conv2d (weight [4C, Cin+C, kh, kw], bias [4C])."""

    def __init__(self, cin: int, cout: int, kernel_size, bias: bool):
        super().__init__()
        kh, kw = kernel_size
        self.weight = nn.Parameter(torch.empty(cout, cin, kh, kw))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        bound = 1.0 / math.sqrt(cin * kh * kw)
        nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            nn.init.uniform_(self.bias, -bound, bound)


class ConvLSTMCell(nn.Module):
    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        self.height, self.width = input_size
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.kernel_size = kernel_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.bias = bias
        if tuple(kernel_size) != (3, 3) or input_dim != hidden_dim or not bias:
            raise NotImplementedError(
                "the HIP ConvLSTM cell covers the stage-4 configuration: 3x3 kernel, input_dim == hidden_dim, bias")
        self.conv = _GateConv(input_dim + hidden_dim, 4 * hidden_dim, kernel_size, bias)

    def forward(self, input, prev_state):
        """One step from an explicit state (src/convLSTM.py:41-56): -> (h_next, c_next), both differentiable."""
        h_cur, c_cur = prev_state
        return ops.convlstm(input.contiguous().unsqueeze(0), self.conv.weight, self.conv.bias, groups=1, return_all=False,
                            state=(h_cur.contiguous(), c_cur.contiguous()))

    def init_hidden(self, batch_size, cuda=True):
        dev = self.conv.weight.device
        z = torch.zeros(batch_size, self.hidden_dim, self.height, self.width, device=dev)
        return (z, z.clone())


class ConvLSTM(nn.Module):
    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, num_layers,
                 batch_first=False, bias=True, return_all_layers=False):
        super().__init__()
        if not (isinstance(kernel_size, tuple) or
                (isinstance(kernel_size, list) and all(isinstance(e, tuple) for e in kernel_size))):
            raise ValueError('`kernel_size` must be tuple or list of tuples')
        kernel_size = kernel_size if isinstance(kernel_size, list) else [kernel_size] * num_layers
        hidden_dim = hidden_dim if isinstance(hidden_dim, list) else [hidden_dim] * num_layers
        if not len(kernel_size) == len(hidden_dim) == num_layers:
            raise ValueError('Inconsistent list length.')
        self.height, self.width = input_size
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.kernel_size = kernel_size
        self.num_layers = num_layers
        self.batch_first = batch_first
        self.bias = bias
        self.return_all_layers = return_all_layers
        cells = []
        for i in range(num_layers):
            cin = input_dim if i == 0 else hidden_dim[i - 1]
            cells.append(ConvLSTMCell((self.height, self.width), cin, hidden_dim[i], kernel_size[i], bias))
        self.cell_list = nn.ModuleList(cells)

    def forward(self, input, hidden_state=None):
        # hidden_state: one (h0, c0) pair per layer, used as given; None is the zero state of get_init_states
        # (src/convLSTM.py:119-120,128).
        if hidden_state is not None and len(hidden_state) != self.num_layers:
            raise ValueError("hidden_state must hold one (h, c) pair per layer")
        # kernel wants (t, b, c, h, w)
        x = input.permute(1, 0, 2, 3, 4) if self.batch_first else input
        layer_outputs, last_states = [], []
        cur = x.contiguous()
        for li, cell in enumerate(self.cell_list):
            st = None if hidden_state is None else (hidden_state[li][0].contiguous(), hidden_state[li][1].contiguous())
            hs, c_last = ops.convlstm(cur, cell.conv.weight, cell.conv.bias, groups=1, return_all=True, state=st)
            layer_outputs.append(hs)
            last_states.append((hs[-1], c_last))
            cur = hs
        out = layer_outputs[-1]
        if self.batch_first:
            out = out.permute(1, 0, 2, 3, 4)
        return out, last_states

    def get_init_states(self, batch_size, cuda=True):
        return [c.init_hidden(batch_size, cuda) for c in self.cell_list]
