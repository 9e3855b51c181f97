"""Randomised shapes / degree patterns for mp_agg_dense_f32 and mp_agg_rows_tiles_f32 against a float64 evaluation: run
boundaries that cut rows, empty tiles, rows past N, rows spanning several waves, F = 64 ... 512, every kernel variant
(MP_FUSED_VARIANT=1 / 3 / 9 in the environment: 32-row / 64-row producer-consumer, one-role kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphgym_amd import ops, _lib
ops.AGG_TILES_MIN_ROWS = 1          # the tile aggregation at every size
from graphgym_amd.graph import CSRGraph
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ.get("SEED", "0")))
def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))
worst = 0.0
for case in range(int(os.environ.get("CASES", "300"))):
    n = ri(1, 400) if case % 3 else (ri(1, 5000) if case % 7 else ri(5000, 60000))
    F = (64, 128, 256, 512)[ri(0, 3)]
    d = 2 * ri(1, 160) if F < 512 or ri(0, 2) else (512, 384, 256)[ri(0, 2)]
    style = ri(0, 4)
    if style == 0:      # uniform sparse
        E = ri(0, 6 * n)
        ei = torch.randint(0, n, (2, E), generator=g)
    elif style == 1:    # a few rows own everything
        E = ri(1, 20 * n)
        hot = torch.randint(0, n, (ri(1, 3),), generator=g)
        ei = torch.stack([hot[torch.randint(0, hot.numel(), (E,), generator=g)], torch.randint(0, n, (E,), generator=g)])
    elif style == 2:    # exactly k entries per row (run boundaries fall on row boundaries for some k)
        k = ri(1, 9)
        ei = torch.stack([torch.arange(n).repeat_interleave(k), torch.randint(0, n, (n * k,), generator=g)])
    elif style == 3:    # only the last rows are populated (leading empty tiles)
        E = ri(1, 4 * n)
        ei = torch.stack([torch.randint(max(0, n - 5), n, (E,), generator=g), torch.randint(0, n, (E,), generator=g)])
    else:               # power-law-ish
        E = ri(1, 10 * n)
        r = (torch.rand(E, generator=g) ** 4 * n).long().clamp(max=n - 1)
        ei = torch.stack([r, torch.randint(0, n, (E,), generator=g)])
    if ei.size(1) == 0:
        continue
    weighted = bool(ri(0, 1))
    w = (torch.rand(ei.size(1), generator=g) + 0.1) if weighted else None
    G = CSRGraph.from_edge_index(ei.to(dev), n, None if w is None else w.to(dev), dst_row=0)
    x = torch.randn(n, F, generator=g).to(dev)
    W = (torch.randn(F, d, generator=g) / F ** 0.5).to(dev)
    b = torch.randn(d, generator=g).to(dev) if ri(0, 1) else None
    relu = bool(ri(0, 1))
    mean = bool(ri(0, 3) == 0)
    s = 0.0 if mean else (0.0, 1.0, 1.5)[ri(0, 2)]
    red = 1 if mean else 0
    if not ops.agg_dense_supported(G, x, W):
        continue
    out, P = ops._raw_agg_dense(G, x, W, b, relu, S=x if s else None, self_scale=s, want_P=True, reduce=red)
    # float64 evaluation of the same sums; errors are measured against the sum of absolute terms of each element
    # (two fp32 summation orders of a 50 000-entry row differ by more than 1e-5 of the result without either being wrong)
    dst, src = ei[0].to(dev), ei[1].to(dev)
    wd = w.to(dev).double().unsqueeze(1) if w is not None else 1.0
    msg = x.double()[src] * wd
    agg = torch.zeros(n, F, dtype=torch.float64, device=dev).index_add_(0, dst, msg)
    mag = torch.zeros(n, F, dtype=torch.float64, device=dev).index_add_(0, dst, msg.abs())
    if mean:
        cnt = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, dst, torch.ones(dst.numel(), dtype=torch.float64, device=dev)).clamp(min=1).unsqueeze(1)
        agg, mag = agg / cnt, mag / cnt
    agg, mag = agg + s * x.double(), mag + abs(s) * x.double().abs()
    pre = agg @ W.double() + (b.double() if b is not None else 0)
    pmag = mag @ W.double().abs() + (b.double().abs() if b is not None else 0)
    ref = torch.relu(pre) if relu else pre
    eP = float(((P.double() - agg).abs() / mag.clamp_min(1.0)).max())
    eO = float(((out.double() - ref).abs() / pmag.clamp_min(1.0)).max())
    eT = 0.0
    if F in ops.AGG_TILES_WIDTHS:       # the aggregation alone on the tile structure
        before = ops.AGG_TILES_CALLS
        yt, _ = ops._raw_spmm(G, x, red, S=x if s else None, self_scale=s)
        assert ops.AGG_TILES_CALLS == before + 1
        eT = float(((yt.double() - agg).abs() / mag.clamp_min(1.0)).max())
    worst = max(worst, eP, eO, eT)
    if eP > 1e-5 or eO > 1e-5 or eT > 1e-5:
        print("MISMATCH", case, n, F, d, style, weighted, relu, mean, s, eP, eO, eT, flush=True)
        sys.exit(1)
print("cases ok, worst relative error", worst)
