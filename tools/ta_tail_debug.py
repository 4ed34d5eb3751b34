#!/usr/bin/env python
"""Debug aid: long-row attention core with the ragged last key tile as rank-1 updates (default) against the same tile swept as a
32-key tile (PRD_TA2_FLAGS=3); prints where the two differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import _lib, ops  # noqa: E402

P, H, c = 64, 4, 16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 449
g = torch.Generator().manual_seed(N)
pair = torch.randn(1, N, N, P, generator=g).cuda()
mask = torch.ones(1, N).cuda()
wts = [(torch.randn(64, P, generator=g) / 8).cuda() for _ in range(4)] + [torch.zeros(64).cuda()]
lib = _lib.lib()
tune0 = lib.prd_get_tune()
outs = []
for tune in (tune0, tune0 | (1 << 6) | (3 << 7)):
    lib.prd_set_tune(tune)
    outs.append(ops.tri_attn_core(pair, mask, wts, H, c, ending=False).clone())
lib.prd_set_tune(tune0)
a, b = outs
d = (a - b)[0]                      # [row, query, 64]
print("rel", float(d.norm() / b.norm()))
print("by channel", [round(float(d[..., k].norm() / b[0][..., k].norm()), 5) for k in range(0, 64, 4)])
print("by query block", [round(float(d[:, q:q + 32].norm() / b[0][:, q:q + 32].norm()), 5) for q in range(0, N, 32)])
r = 5
print("row 5 query 7 head 0:", a[0, r, 7, :16].tolist(), b[0, r, 7, :16].tolist())
print("ratio:", (a[0, r, 7, :16] / b[0, r, 7, :16]).tolist())


def run(mask_):
    res = []
    for tune in (tune0, tune0 | (1 << 6) | (3 << 7)):
        lib.prd_set_tune(tune)
        res.append(ops.tri_attn_core(pair, mask_, wts, H, c, ending=False).clone())
    lib.prd_set_tune(tune0)
    return res


m1 = mask.clone(); m1[0, N - 1] = 0
a, b = run(m1)
print("tail key masked: rel", float((a - b).norm() / b.norm()))
m2 = torch.zeros_like(mask); m2[0, N - 1] = 1; m2[0, 3] = 1
a, b = run(m2)
print("only keys 3 and tail valid: row N-1 rel", float((a[0, N - 1] - b[0, N - 1]).norm() / b[0, N - 1].norm()))
print(" query 3 of row 3, head 0:", a[0, 3, 3, :8].tolist(), b[0, 3, 3, :8].tolist())
m3 = torch.zeros_like(mask); m3[0, N - 1] = 1
a, b = run(m3)
print("only the tail key valid: row N-1 rel", float((a[0, N - 1] - b[0, N - 1]).norm() / b[0, N - 1].norm()))
print(" query 3, head 0:", a[0, N - 1, 3, :8].tolist(), b[0, N - 1, 3, :8].tolist())

# host-side reference for row 3 with keys {3, N-1}: which factor is off?
import torch.nn.functional as F
x = F.layer_norm(pair[0, 3], (P,))                      # [N, P]
wq, wk, wv, wg, bg = wts
h = 0
q = (x @ wq[h * 16:(h + 1) * 16].T) * 0.25
k = x @ wk[h * 16:(h + 1) * 16].T
v = x @ wv[h * 16:(h + 1) * 16].T
gt = torch.sigmoid(x @ wg[h * 16:(h + 1) * 16].T + bg[h * 16:(h + 1) * 16])
qi = 3
s3, st = float(q[qi] @ k[3]), float(q[qi] @ k[N - 1])
w3 = 1.0 / (1.0 + torch.exp(torch.tensor(st - s3)))
print("logits", s3, st, "w3", float(w3))
ref = gt[qi] * (w3 * v[3] + (1 - w3) * v[N - 1])
m2 = torch.zeros_like(mask); m2[0, N - 1] = 1; m2[0, 3] = 1
a, b = run(m2)
print("ref ", ref[:8].tolist())
print("a   ", a[0, 3, qi, :8].tolist())
print("b   ", b[0, 3, qi, :8].tolist())
# solve the weight a used, per channel: a = g (w v3 + (1-w) vt)
wa = (a[0, 3, qi, :16] / gt[qi] - v[N - 1]) / (v[3] - v[N - 1])
print("weight implied by a, per channel:", wa.tolist())

qr, kt = (x[qi] @ wq[:16].T), k[N - 1]
print("target q.k (unscaled) ~ 0.77924; true", float(qr @ kt))
import itertools
sets = {"c0-7": list(range(8)), "c8-15": list(range(8, 16)), "half0": [0, 1, 2, 3, 8, 9, 10, 11], "half1": [4, 5, 6, 7, 12, 13, 14, 15]}
for n1, s1 in sets.items():
    print(n1, "sum", float(qr[s1] @ kt[s1]), " x2", 2 * float(qr[s1] @ kt[s1]))
for n1, s1 in sets.items():
    for n2, s2 in sets.items():
        if n1 != n2:
            print("q", n1, "k", n2, float(qr[s1] @ kt[s2]), " +swap", float(qr[s1] @ kt[s2] + qr[s2] @ kt[s1]))
# other keys of the last block standing in for the tail key?
for pos in (N - 1, 448 - 32, 447, 0, 3):
    print("q . k[%d]" % pos, float(qr @ k[pos]))
