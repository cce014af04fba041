"""Time the label-tile generator alone: python tools/gen_time.py N T P [reps]  (events around reps launches)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from scoary_amd.engine import AssociationEngine, pack_bits_rows

def main():
    N, T, P = (int(x) for x in sys.argv[1:4])
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    frac = float(os.environ.get("GEN_FRAC", "0.3"))
    eng = AssociationEngine()
    rng = np.random.default_rng(1)
    traits = (rng.random((T, N)) < frac).astype(np.uint8)
    tb, mb = pack_bits_rows(traits), pack_bits_rows(np.ones((T, N), np.uint8))
    masks, trv = eng.vecrows(mb, N), eng.vecrows(tb, N)
    _, margins = eng.counts(eng.pack_dense(np.ones((1, N), dtype=np.uint8)), trv, masks)
    out = eng.perm_generate_tiles(masks, margins, N, P, 0, 7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            eng.perm_generate_tiles(masks, margins, N, P, 0, 7, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    print("N=%d T=%d P=%d frac=%.2f debug=%s: %.4f ms per launch" % (N, T, P, frac, os.environ.get("SCOARY_LABELS_DEBUG", "0"), best))

main()
