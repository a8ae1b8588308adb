"""pagg_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by pathnet_amd/).

CPU restatement (torch CPU tensors, explicit index formulas, explicit recurrent cell) of the
reference path aggregator forward, for the three classes that share the
``forward(X, neis, num_w, walk_len, indices, layer_type, indxx)`` surface:

  variant "hetero" : class PathNet       /root/reference/PathNet_run.py:150-211
  variant "homo"   : class PathNet_homo  /root/reference/PathNet_run.py:214-278
  variant "pagg"   : class PAGG          /root/reference/baseline/GPRGNN/src/copy.py:299-359

Parity pinning: the reference ships no numeric golden vectors for this path.
This restatement is pinned (a) in this container against the reference classes themselves,
loaded with ``ast`` from /root/reference (tests/test_oracle_pagg.py, tests/ref_extract.py) and
(b) everywhere against tests/golden/pagg_*.npz, which tests/golden/make_golden_pagg.py produced by
running those reference classes.

The restatement is written as an index *plan* + dense math so that it documents the reference's
index quirks explicitly (they are reference behaviour and are reproduced, not fixed):

  P = S*W paths, flat path p = s*W + w, ids/codes are [P, L].
  homo / pagg  (PathNet_run.py:246-267, copy.py:334-351)
      slot q = p, step t      : node = ids[q, t],            code = codes[q, t]
      pooling group of slot q : q // W  (member q % W)
  hetero       (PathNet_run.py:179-204)
      rows are built time-major and reversed (flip, :182-183), the codes are read path-major
      (:184), and the result is re-viewed as [P, L] (:191-192):
      slot q, step t, r = q*L + t : node = ids[r % P, L-1 - r // P],  code = codes[q, t]
      the final hidden states are viewed [W, S, H] (:196-197):
      pooling group of slot q : q % S   (member q // S)
      attention ego of slot q : Xh[ids[q, 0]]   (neis[0].view(W, S, H), :199)
"""
import numpy as np
import torch

VARIANTS = ("hetero", "homo", "pagg")


def plan(variant, ids, codes, S, W, L):
    """ids, codes: integer arrays [S*W, L].  Returns (node[P,L], code[P,L], group[P], member[P], ego[P])
    as int64 numpy arrays, indexed by sequence slot q."""
    ids = np.asarray(ids).reshape(S * W, L).astype(np.int64)
    codes = np.asarray(codes).reshape(S * W, L).astype(np.int64)
    P = S * W
    q = np.arange(P, dtype=np.int64)
    if variant in ("homo", "pagg"):
        node, code = ids.copy(), codes.copy()
        group, member = q // W, q % W
    elif variant == "hetero":
        r = q[:, None] * L + np.arange(L, dtype=np.int64)[None, :]
        node = ids[r % P, L - 1 - r // P]
        code = codes.copy()
        group, member = q % S, q // S
    else:
        raise ValueError(variant)
    ego = ids[:, 0].copy()
    return node, code, group, member, ego


def _lin(x, w, b):
    return x @ w.t() + b


def project(variant, fc0_w, fc0_b, X):
    """fc0 (+ ReLU for the homo class): PathNet_run.py:175 / :242-243 / copy.py:330."""
    Xh = _lin(X, fc0_w, fc0_b)
    return torch.relu(Xh) if variant == "homo" else Xh


def forward(variant, params, X, ids, codes, sel, W, L, drop_seq=None, drop_cls=None, dtype=torch.float32,
            return_intermediates=False, Xh=None, cell=None):
    """Reference forward, restated.

    params : dict name -> tensor with the reference state_dict keys (fc0.*, nets.<d>.* or nei<d>.*,
             LSTM.* or RNN.*, attw.*, fc2.*)
    X      : [N, F];  ids, codes: [S, W, L] (or [S*W, L]) integers;  sel: [S] indices of the masked nodes
    drop_seq : None or multiplicative mask [L, P, H] already scaled by 1/(1-p)  (F.dropout on the
               recurrent input, PathNet_run.py:194 / :264 / copy.py:348)
    drop_cls : None or mask [S, 2H] (F.dropout on the classifier input, :209 / :276 / copy.py:357)
    cell     : None = the class's own path encoder (nn.LSTM; nn.RNN for PAGG).  "gru" / "mean" / "sum" are the ablation
               rows of the paper's table, which the reference ships no code for (README.md:118): "gru" is torch.nn.GRU's
               cell (parameters GRU.*, gate order r, z, n), "mean" / "sum" the mean / sum over the L steps of the
               dropped-out step rows (no recurrent parameters); "lstm" / "rnn" force those cells for any class.
    """
    g = {k: (v.detach() if not v.requires_grad else v).to(dtype) for k, v in params.items()}
    S = len(sel)
    P = S * W
    node, code, group, member, ego = plan(variant, np.asarray(ids), np.asarray(codes), S, W, L)
    H = g["fc2.weight"].shape[1] // 2

    if Xh is None:
        Xh = project(variant, g["fc0.weight"], g["fc0.bias"], X.to(dtype))
    else:
        Xh = Xh.to(dtype)       # already projected (node-sharded runs project row blocks separately)

    # distance-indexed Linear bank: row (q, t) goes through nets[code[q, t]] only (:249-255)
    nd = torch.from_numpy(node.reshape(-1))
    cd = torch.from_numpy(code.reshape(-1))
    rows = Xh[nd]                                                       # the gather, :179 / :246
    y = torch.zeros(P * L, H, dtype=dtype)
    for d in range(L):
        wname = ("nei%d" % d) if variant == "pagg" else ("nets.%d" % d)
        m = cd == d
        if m.any():
            y[m] = _lin(rows[m], g[wname + ".weight"], g[wname + ".bias"])
    y = y.view(P, L, H)
    if variant == "homo":
        y = torch.relu(y)                                               # :257
    xs = y.transpose(0, 1)                                              # [L, P, H]
    if drop_seq is not None:
        xs = xs * drop_seq.to(dtype)

    # recurrent cell over the L steps, zero initial state, only the final h is used
    cell = cell or ("rnn" if variant == "pagg" else "lstm")
    if cell in ("mean", "sum"):
        h = xs.sum(dim=0) if cell == "sum" else xs.mean(dim=0)
    elif cell == "gru":
        wih, whh = g["GRU.weight_ih_l0"], g["GRU.weight_hh_l0"]
        bih, bhh = g["GRU.bias_ih_l0"], g["GRU.bias_hh_l0"]
        h = torch.zeros(P, H, dtype=dtype)
        for t in range(L):
            gi = xs[t] @ wih.t() + bih
            gh = h @ whh.t() + bhh
            i_r, i_z, i_n = gi.split(H, dim=1)
            h_r, h_z, h_n = gh.split(H, dim=1)
            r = torch.sigmoid(i_r + h_r)
            z = torch.sigmoid(i_z + h_z)
            n = torch.tanh(i_n + r * h_n)
            h = (1.0 - z) * n + z * h
    elif cell == "rnn":
        wih, whh = g["RNN.weight_ih_l0"], g["RNN.weight_hh_l0"]
        b = g["RNN.bias_ih_l0"] + g["RNN.bias_hh_l0"]
        h = torch.zeros(P, H, dtype=dtype)
        for t in range(L):
            h = torch.tanh(xs[t] @ wih.t() + h @ whh.t() + b)
    elif cell == "lstm":
        wih, whh = g["LSTM.weight_ih_l0"], g["LSTM.weight_hh_l0"]
        b = g["LSTM.bias_ih_l0"] + g["LSTM.bias_hh_l0"]
        h = torch.zeros(P, H, dtype=dtype)
        c = torch.zeros(P, H, dtype=dtype)
        for t in range(L):
            gates = xs[t] @ wih.t() + h @ whh.t() + b
            i, f, gg, o = gates.split(H, dim=1)                         # torch gate order i, f, g, o
            i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
            c = f * c + i * gg
            h = o * torch.tanh(c)
    else:
        raise ValueError("unknown cell %r" % (cell,))

    # pooling over the W members of each group
    order = torch.from_numpy(np.argsort(group * W + member, kind="stable"))
    hg = h[order].view(S, W, H)                                         # [group, member, H]
    if variant == "homo":
        ego_rows = y[:, 0, :][order].view(S, W, H)                      # ego_full, :259-260
        att = _lin(torch.cat([hg, ego_rows], dim=-1), g["attw.weight"], g["attw.bias"])
        pooled = ((1.0 + att) * hg).mean(dim=1)                         # :269-273
    elif variant == "hetero":
        ego_rows = Xh[torch.from_numpy(ego)][order].view(S, W, H)       # neis[0].view(W,S,H), :199
        sc = _lin(torch.cat([hg, ego_rows], dim=-1), g["attw.weight"], g["attw.bias"])
        sc = torch.nn.functional.leaky_relu(sc, 0.01)
        att = torch.softmax(sc, dim=1)                                  # implicit dim 0 of [W,S,1], :200-201
        pooled = (att * hg).mean(dim=1)                                 # :202-204
    else:
        att = None
        pooled = hg.mean(dim=1)                                         # copy.py:353

    layer1 = torch.cat([Xh[torch.as_tensor(np.asarray(sel), dtype=torch.long)], pooled], dim=1)
    if drop_cls is not None:
        layer1 = layer1 * drop_cls.to(dtype)
    out = _lin(layer1, g["fc2.weight"], g["fc2.bias"])
    if return_intermediates:
        return out, {"Xh": Xh, "y": y, "h": h, "pooled": pooled, "att": att}
    return out
