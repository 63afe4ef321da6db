"""One-off campaign (run by hand on an MI355X): production kernel vs the plain-C oracle (oracle/invert_c.c, an independent
CPU implementation of the reference's kernel) on 1e7 pixels of the benchmark scene, mono and dual-pol, identical dB
inputs: indices and NaN masks must be equal."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import cport  # noqa: E402
from oracle import invert as oinv  # noqa: E402
from oracle import lut as olut  # noqa: E402
from xsarsea_amd import _lib  # noqa: E402
from xsarsea_amd.windspeed import _engine, get_model  # noqa: E402

n = int(os.environ.get("XSW_CAMPAIGN_N", "3200"))
res = os.environ.get("XSW_CAMPAIGN_RES", "high")  # "low": the 51x250x73 / 51x386 LUTs
kw = {} if res == "high" else {"resolution": "low"}
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
lco, lcr = get_model("gmf_cmod5n")._lut(units="dB", **kw), get_model("gmf_s1_v2")._lut(units="dB", **kw)
ctx.upload_luts(co=_engine._co_dict(lco), cr=_engine._cr_dict(lcr))
inc, s_vv, anc = bench.make_scene(n, n, 20000, 8000, int(os.environ.get("XSW_CAMPAIGN_SEED", "77")), dev)
inc, s_vv, anc = (t.cpu().numpy() for t in (inc, s_vv, anc))
rng = np.random.default_rng(7)
s_vh = (s_vv * 0.02 * rng.gamma(100, 1 / 100, s_vv.shape) + 10 ** -3.5).astype(np.float32)
dsig = ((1.25 / (s_vh / 10 ** -3.5)) ** 4).astype(np.float32)
sco, scr = oinv.to_db(s_vv), oinv.to_db(s_vh)  # numpy float32 dB, fed to both sides
prep = oinv.Prepared(olut.to_lut("gmf_cmod5n", **kw), olut.to_lut("gmf_s1_v2", **kw))
bad = 0
for mode in ("mono", "dual"):
    cr_in = (scr, dsig) if mode == "dual" else (None, None)
    t = time.perf_counter()
    nanr = np.full(inc.shape, np.nan, dtype=np.float32)
    o = cport.invert_numpy(prep, inc, sco, cr_in[0] if mode == "dual" else nanr, cr_in[1] if mode == "dual" else nanr, anc,
                           return_idx=True, reference_layout=False)
    t_cpu = time.perf_counter() - t
    g = ctx.invert_host(inc, sigma0_co=sco, sigma0_cr=cr_in[0], dsig_cr=cr_in[1], anc=anc, sigma0_is_db=True, algo="pruned",
                        want_idx=True)
    cols = slice(0, 3) if mode == "dual" else slice(0, 2)
    d = int((g[2][..., cols] != o[2][..., cols]).any(axis=-1).sum())
    m = int((np.isnan(g[0]) != np.isnan(o[0])).sum())
    bad += d + m
    print(f"{mode}: {n}x{n} px, index mismatches = {d}, NaN-mask mismatches = {m}, C oracle {t_cpu:.0f} s on {cport.max_threads()} threads",
          flush=True)
print("TOTAL mismatches:", bad)
