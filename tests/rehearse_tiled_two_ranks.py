"""`multi_gpu.invert_from_model_tiled` with REAL inversions, several ranks on ONE MI355X (gloo gather through host memory;
RCCL needs two devices), compared on rank 0 with the single-process call on the full raster.  The parent never touches the
GPU (its children are started by spawn).  Run by tests/test_gpu_configs.py::test_tiled_api_real_inversions (3 ranks), or by hand:
    python tests/rehearse_tiled_two_ranks.py [ranks]"""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        import xsarsea_amd
        from xsarsea_amd import multi_gpu, windspeed
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_kernel import synthetic_scene
        xsarsea_amd.options.device = 0
        warnings.simplefilter("ignore")
        inc, s_vv, s_vh, dsig, anc = synthetic_scene(301, 517, np.float32, 9)
        anc[:100] = np.nan  # rank 0's whole tile (3 ranks: lines 0..99) has no ancillary wind: NaN rows, no per-tile assertion
        mono = multi_gpu.invert_from_model_tiled(inc, s_vv, ancillary_wind=anc, model="gmf_cmod5n")
        dual = multi_gpu.invert_from_model_tiled(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=dsig, model=("gmf_cmod5n", "gmf_s1_v2"))
        if rank == 0:
            ref_m = windspeed.invert_from_model(inc, s_vv, ancillary_wind=anc, model="gmf_cmod5n")
            ref_d = windspeed.invert_from_model(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=dsig, model=("gmf_cmod5n", "gmf_s1_v2"))
            eq = lambda a, b: bool(np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64)))
            ret["mono"] = eq(mono, ref_m)
            ret["dual"] = eq(dual[0], ref_d[0]) and eq(dual[1], ref_d[1])
            ret["shape"] = tuple(np.shape(mono))
            ret["nan_tile"] = bool(np.isnan(mono[:100]).all() and not np.isnan(mono[100:]).all())
        else:
            assert mono is None and dual is None
        # a 1-D incidence row with a SQUARE raster: passed whole to every tile
        sq = 96
        sq_mono = multi_gpu.invert_from_model_tiled(inc[0, :sq], s_vv[:sq, :sq], ancillary_wind=anc[150:150 + sq, :sq], model="gmf_cmod5n",
                                                    resolution="low")
        if rank == 0:
            ref = windspeed.invert_from_model(inc[0, :sq], s_vv[:sq, :sq], ancillary_wind=anc[150:150 + sq, :sq], model="gmf_cmod5n",
                                              resolution="low")
            ret["square_1d_inc"] = eq(sq_mono, ref)
        # a raster of a few lines: chunks thinner than the staging ring's minimum (numpy's dB of those rows is uploaded instead)
        thin = multi_gpu.invert_from_model_tiled(inc[:7], s_vv[:7], s_vh[:7], ancillary_wind=anc[150:157], dsig_cr=0.2,
                                                 model=("gmf_cmod5n", "gmf_s1_v2"), resolution="low")
        if rank == 0:
            ref = windspeed.invert_from_model(inc[:7], s_vv[:7], s_vh[:7], ancillary_wind=anc[150:157], dsig_cr=0.2,
                                              model=("gmf_cmod5n", "gmf_s1_v2"), resolution="low")
            ret["thin"] = eq(thin[0], ref[0]) and eq(thin[1], ref[1])
        # DEVICE tensors in -> torch tensors out on rank 0 (round 5: the gathered call takes CUDA tensors; codes stay on the device)
        import torch
        dev = torch.device("cuda", 0)
        t_inc, t_s, t_vh, t_dsig, t_anc = (torch.from_numpy(a).to(dev) for a in (inc, s_vv, s_vh, dsig, anc))
        d_mono = multi_gpu.invert_from_model_tiled(t_inc, t_s, ancillary_wind=t_anc, model="gmf_cmod5n")
        d_dual = multi_gpu.invert_from_model_tiled(t_inc, t_s, t_vh, ancillary_wind=t_anc, dsig_cr=t_dsig, model=("gmf_cmod5n", "gmf_s1_v2"), n_chunks=3)
        d_cross = multi_gpu.invert_from_model_tiled(t_inc, t_vh, dsig_cr=0.3, model="gmf_s1_v2")
        if rank == 0:
            r_mono = windspeed.invert_from_model(t_inc, t_s, ancillary_wind=t_anc, model="gmf_cmod5n")
            r_dual = windspeed.invert_from_model(t_inc, t_s, t_vh, ancillary_wind=t_anc, dsig_cr=t_dsig, model=("gmf_cmod5n", "gmf_s1_v2"))
            r_cross = windspeed.invert_from_model(t_inc, t_vh, dsig_cr=0.3, model="gmf_s1_v2")
            torch.cuda.synchronize()
            teq = lambda a, b: bool(torch.is_tensor(a) and a.is_cuda and a.shape == b.shape and
                                    torch.equal(torch.view_as_real(a).view(torch.int64), torch.view_as_real(b).view(torch.int64)))
            ret["device_mono"] = teq(d_mono, r_mono)
            ret["device_dual"] = teq(d_dual[0], r_dual[0]) and teq(d_dual[1], r_dual[1])
            ret["device_cross"] = bool(torch.equal(d_cross.view(torch.int64), r_cross.view(torch.int64)))
        else:
            assert d_mono is None and d_dual is None and d_cross is None
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with mp.Manager() as m:
        ret = m.dict()
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        mp.spawn(worker, args=(n, port, ret), nprocs=n, join=True)
        print(f"invert_from_model_tiled, {n} ranks on one device (gloo): raster", ret.get("shape"), "mono bit-equal to the single-process call:",
              ret.get("mono"), "dual:", ret.get("dual"), "all-NaN-ancillary tile:", ret.get("nan_tile"), "1-D incidence on a square raster:",
              ret.get("square_1d_inc"))
        print("thin raster:", ret.get("thin"), "device tensors mono / dual / cross-only:", ret.get("device_mono"), ret.get("device_dual"), ret.get("device_cross"))
        sys.exit(0 if all(ret.get(k) for k in ("mono", "dual", "nan_tile", "square_1d_inc", "thin", "device_mono", "device_dual", "device_cross")) else 1)
