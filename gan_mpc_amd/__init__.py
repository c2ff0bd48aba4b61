"""gan_mpc_amd: the GAN-MPC inner loop of returaj/gan_mpc as hand-written HIP kernels for MI355X
(gfx950) behind a C ABI (include/gan_mpc_amd.h), with a Python host layer that mirrors the
reference's model / policy / trainer protocol."""

from ._lib import GmpcError, LIB_PATH, load  # noqa: F401

__all__ = ["GmpcError", "LIB_PATH", "load"]
