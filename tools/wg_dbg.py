import torch, sys
sys.path.insert(0, '.')
from protein_redesign_amd import ops
for rows, O, I in [(102400, 256, 64), (20000, 64, 256), (9001, 128, 128), (8192, 256, 256), (102400, 64, 64)]:
    g = torch.Generator().manual_seed(rows + O)
    wide = torch.randn(rows, O + 64, generator=g).cuda()
    dy = wide[:, 64:]
    x = torch.randn(rows, I, generator=g).cuda()
    got = ops.linear_wgrad(dy, x)
    want = dy.double().t() @ x.double()
    ref32 = dy.t() @ x
    e1 = ((got.double() - want).norm() / want.norm()).item()
    e2 = ((ref32.double() - want).norm() / want.norm()).item()
    d = (got.double() - want).abs()
    print(rows, O, I, "kernel err %.3e  torch fp32 err %.3e  max abs %.3e at" % (e1, e2, d.max().item()), divmod(int(d.argmax()), I))
