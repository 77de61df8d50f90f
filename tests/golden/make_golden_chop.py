"""Golden fixture G23: the reference's tiled whole-image evaluation (`forward_chop`) - pins oracle.sr_oracle.forward_chop.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_chop.py
The REAL method ContrastiveBlindQEDSRHandler.forward_chop (rumpy/SISR/models/blur_kernel_blind_sr/handlers.py:907-945; SANHandler's,
advanced/handlers.py:85-123, is the same text with `super().run_eval(chunk)[0]` for `self.run_chopped_eval(chunk)`) is called on a stand-in
object that carries what the method reads (`max_combined_im_size`, `scale`, `run_chopped_eval`, itself for the recursion).  The per-tile
"network" is a fixed function of the tile alone - every pixel repeated scale x scale times plus a term that depends on the position INSIDE
the tile - so a stitching offset, a wrong overlap or a wrong recursion threshold changes the result.  Two cases: one level of recursion
(61 x 83, threshold 1500) and none (threshold 10**6); a third at scale 4 with a ragged size."""
import os
import runpy
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
shim = runpy.run_path(os.path.join(HERE, 'make_golden.py'), run_name='shim_only')
from rumpy.SISR.models.blur_kernel_blind_sr.handlers import ContrastiveBlindQEDSRHandler  # noqa: E402


def tile_function(scale):
    """what the stand-in 'network' computes for one tile (also used by tests/test_oracle_golden.py)"""
    def run(chunk):
        up = chunk.repeat_interleave(scale, dim=2).repeat_interleave(scale, dim=3)
        h, w = up.shape[2], up.shape[3]
        pos = (torch.arange(h, dtype=torch.float32).view(1, 1, h, 1) * 0.001953125 + torch.arange(w, dtype=torch.float32).view(1, 1, 1, w) * 0.00048828125)
        return up + pos
    return run


CASES = [('recursive', 2, (61, 83), 1500), ('flat', 2, (61, 83), 10 ** 6), ('x4', 4, (37, 50), 600)]


def main():
    d = {}
    for tag, scale, (h, w), limit in CASES:
        g = torch.Generator().manual_seed(2300 + h)
        x = torch.rand(1, 2, h, w, generator=g)
        stub = types.SimpleNamespace(max_combined_im_size=limit, scale=scale, run_chopped_eval=tile_function(scale))
        stub.forward_chop = types.MethodType(ContrastiveBlindQEDSRHandler.forward_chop, stub)
        out = stub.forward_chop(x)
        d[tag + '_x'], d[tag + '_out'] = x.numpy(), out.numpy()
        d[tag + '_meta'] = np.asarray([scale, limit])
    np.savez_compressed(os.path.join(HERE, 'g23_forward_chop.npz'), **d)
    print('wrote g23', {k: v.shape for k, v in d.items()})


if __name__ == '__main__':
    main()
