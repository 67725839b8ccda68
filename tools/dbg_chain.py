import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_hip_parity import _run_basicvsr, rel_l2
dev = torch.device("cuda:0")
for shape, blocks in [((1, 1, 3, 20, 40), 1), ((1, 1, 3, 64, 64), 1), ((1, 1, 3, 64, 64), 2), ((1, 3, 3, 20, 40), 3)]:
    out = {}
    for mode in ("1", "0", "2", "1b"):
        os.environ["VSRLAB_AMD_CHAIN"] = mode[0]
        _, _, _, sr, grads = _run_basicvsr("bf16", 64, blocks, shape, 93, 94, dev)
        out[mode] = (sr, grads)
    for m in ("1", "2", "1b"):
        d = (out["0"][0] - out[m][0]).abs()
        print(shape, blocks, m, "sr: differing", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()), "rel", rel_l2(out[m][0], out["0"][0]))
        bad = [(k, rel_l2(out[m][1][k], v)) for k, v in out["0"][1].items() if not torch.equal(out[m][1][k], v)]
        print("  grads differing:", len(bad), "of", len(out["0"][1]), bad[:3])
