"""CPU oracle for the xsarsea wind-inversion hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy + plain C)
of the reference algorithm (umr-lops/xsarsea: `invert_from_model`,
`sigma0_detrend`, the GMF formulas and the LUT build).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it, and only as the checker / reported baseline.  The product package
`xsarsea_amd` never imports anything from here.

Parity pinning (see DESIGN.md "Oracle"):
  * The per-pixel search kernel restatement (`oracle.invert`) is pinned
    bit-for-bit against golden vectors produced by executing the reference's
    own kernel body (`src/xsarsea/windspeed/windspeed.py:183-282`) in the
    build container (`tests/golden/make_golden.py`, fixtures in
    `tests/golden/*.npz`).
  * The GMF restatements (`oracle.gmf`) are pinned against the reference's
    scalar GMF functions evaluated on a lattice (fixture `gmf_lattice.npz`)
    and against the only known-answer values the reference holds
    (`gmfs.py:60-63`).
  * The LUT interpolation (`oracle.lut`) follows xarray's orthogonal 1-D
    decomposition with scipy `interp1d`; xarray itself is absent from the
    image so bit-equality with the reference's `DataArray.interp` is
    UNPINNED (formula-pinned only).
  * The cross-pol preprocessing (`oracle.crosspol`: get_dsig, get_dsig_wspd,
    nesz_flattening) is pinned bit-for-bit against `crosspol_prep.npz`,
    produced by executing the reference's `windspeed/utils.py`.
"""
