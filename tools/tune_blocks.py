"""Sweep of SV_OPT_PERSISTENT_BLOCKS (the block budget of conv3x3p / wgrad3x3) on the WRN-28-2 body shapes at the
grouped step's size (4 x 512 images): python tools/tune_blocks.py"""
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from shot_vae_amd import _lib as L          # noqa: E402
from tools.layer_bench import bench_layer   # noqa: E402

for (Cin, H) in ((32, 32), (64, 16), (128, 8)):
    for blocks in (256, 512, 768, 1024, 1536, 2048):
        with L.options(persistent_blocks=blocks):
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                r = bench_layer(2048, Cin, H, Cin)
            print("C=%3d blocks=%4d  fwd %7.1f  dgrad %7.1f  wgrad %7.1f us" % (Cin, blocks, r["fwd"], r["dgrad"], r["wgrad"]),
                  flush=True)
