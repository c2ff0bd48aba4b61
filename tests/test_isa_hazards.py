"""The ISA of every kernel of the library, checked for the MFMA -> vector read hazard hipcc under-pads on gfx950
(tests/repro/README.md section 2, tests/repro/check_mfma_hazards.py).  Cross-compiles the device code of each unit
to a listing (no GPU needed; listings are cached next to the sources, keyed by modification time)."""

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gan_mpc_amd", "csrc")
ASM = os.path.join(CSRC, "asm")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, os.path.join(ROOT, "tests", "repro"))
import check_mfma_hazards as chk  # noqa: E402


def _listing(src):
    out = os.path.join(ASM, os.path.basename(src)[:-4] + ".s")
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "gan_mpc_amd.h")]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-w",
                           "--cuda-device-only", "-S", src, "-o", out], cwd=CSRC,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_no_kernel_reads_an_mfma_destination_too_early():
    os.makedirs(ASM, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert len(srcs) >= 15
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        listings = list(ex.map(_listing, srcs))
    bad = [b for f in listings for b in chk.check(f)]
    assert not bad, "\n".join(f"{os.path.basename(p)}:{l1}: {name}: `{text}` {ws} wait states after {op} "
                              f"(line {l0}), {need} needed" for p, name, l0, op, l1, text, ws, need in bad[:20])


def test_the_checker_flags_the_known_cases():
    rep = os.path.join(ROOT, "tests", "repro")
    hinted = chk.check(os.path.join(rep, "linearize_nt2_hinted.s"))
    # the epilogue's first read, a31, 12 wait states after the k-loop's last MFMA (the 3.8e-2 error of round 2)
    assert any(l0 == 1634 and l1 == 1647 and ws == 12 and need == 18 for _, _, l0, _, l1, _, ws, need in hinted)
    default = chk.check(os.path.join(rep, "linearize_nt2_default.s"))
    assert any(l0 == 973 and l1 == 991 for _, _, l0, _, l1, _, _, _ in default)
    assert not any(l1 == 1647 for _, _, _, _, l1, _, _, _ in default)
