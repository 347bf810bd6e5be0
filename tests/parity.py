"""Comparison helpers shared by the parity tests."""
import torch


def codes_match_up_to_near_ties(codes: torch.Tensor, ref: torch.Tensor, rel_gap: torch.Tensor, tol: float = 2e-5) -> int:
    """RVQ code comparison that knows about near ties: the first level (inside its residual group: level 0 alone, levels 1..7
    chained) at which a frame differs from the reference must be a decision whose top-2 distance gap the fixture recorded as
    below ``tol`` (the latent behind it carries ~1e-6 of fp32 summation-order noise); anything else is a failure.  Returns the
    number of frames excused that way."""
    excused = 0
    B, K, Fr = ref.shape
    for b in range(B):
        for f in range(Fr):
            for lo, hi in ((0, 1), (1, K)):
                d = (codes[b, lo:hi, f] != ref[b, lo:hi, f]).nonzero()
                if d.numel():
                    lvl = lo + int(d[0])
                    assert float(rel_gap[b, lvl, f]) < tol, f"code mismatch at b={b} level={lvl} frame={f}: gap {float(rel_gap[b, lvl, f]):.2e}"
                    excused += 1
    return excused
