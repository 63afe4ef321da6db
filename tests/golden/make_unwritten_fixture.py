"""Run with the build container's /opt/conda/bin/python3.9 (h5py).  Two small HDF5 files with storage that was never written (tests of hdf5_min's fill value): a chunked dataset of which only some
chunks exist (fill value message: -999.0; and one with no fill value defined), and a contiguous dataset that was never written."""
import os, h5py, numpy as np
HERE = "/root/repo/tests/golden/nc4"
path = os.path.join(HERE, "unwritten_storage.h5")
with h5py.File(path, "w", libver="earliest") as f:
    d = f.create_dataset("partly", shape=(6, 8), dtype="<f8", chunks=(3, 4), fillvalue=-999.0)
    d[0:3, 0:4] = np.arange(12.0).reshape(3, 4)
    d2 = f.create_dataset("partly_nofill", shape=(6, 8), dtype="<f4", chunks=(3, 4), compression="gzip")
    d2[3:6, 4:8] = np.arange(12.0, dtype="<f4").reshape(3, 4)
    f.create_dataset("never", shape=(4, 5), dtype="<f8", fillvalue=7.5)
    f.create_dataset("never_int", shape=(3,), dtype="<i4")
exp = {}
with h5py.File(path, "r") as f:
    for k in f:
        exp[k] = f[k][...]
np.savez(path + ".expected.npz", **exp)
print({k: v.tolist() for k, v in exp.items() if k != "partly"})
