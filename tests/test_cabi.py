"""The C-ABI library loads (even without a GPU) and exports every symbol include/tidehunter_amd.h
declares; compute entry points fail loudly when there is no device."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tidehunter_amd", "libtidehunter_amd.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(SO):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tidehunter_amd", "csrc")], check=True)
    return C.CDLL(SO)


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "tidehunter_amd.h")).read()
    names = set(re.findall(r"\b(th_[a-z_]+)\s*\(", hdr))
    assert {"th_init", "th_process_batch", "th_batch_upload", "th_batch_run", "th_batch_collect", "th_output", "th_run_file",
            "th_init_para", "th_alloc_tandem_seq", "th_destroy"} <= names
    for n in sorted(names):
        assert hasattr(lib, n), n


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import tidehunter_amd as TH
    with pytest.raises(RuntimeError, match="no HIP device"):
        TH.TideHunter(device=0)


def test_product_does_not_link_oracle_or_emulation():
    out = subprocess.run(["nm", "-D", "--defined-only", SO], capture_output=True, text=True).stdout
    assert "tho_" not in out and "hipemul" not in out
    ldd = subprocess.run(["ldd", SO], capture_output=True, text=True).stdout
    assert "libtho" not in ldd and "libth_emul" not in ldd


def test_synth_reads_are_deterministic(lib):
    import tidehunter_amd as TH
    l = TH.load_library()
    a = TH.synth_read(l, 1, 5, 1000, 10, 0.1); b = TH.synth_read(l, 1, 5, 1000, 10, 0.1); c = TH.synth_read(l, 1, 6, 1000, 10, 0.1)
    assert a == b and a != c and 9000 < len(a) < 11000 and set(a) <= set(b"ACGT")


def test_isa_untracked_lds_read_is_not_touched_before_its_wait(tmp_path):
    """th_k_chain_dp requests its next trip's records with an LDS read the compiler does not track (th_hw.h: TH_LDS_READ64_ASYNC /
    TH_LDS_ASYNC_WAIT); tools/isa_check_async_lds.py looks at the gfx950 assembly of the product build for any instruction that touches
    the destination registers between the read and its wait"""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "tidehunter_amd", "csrc")
    out = tmp_path / "th_pipeline.s"
    flags = subprocess.run(["make", "-s", "--no-print-directory", "-C", src, "print-hipflags"], capture_output=True, text=True, check=True).stdout.split()      # the product's own
    assert "--offload-arch=gfx950" in flags
    subprocess.run([hipcc] + flags + ["-w", "-I" + src, "-I" + os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", os.path.join(src, "th_pipeline.hip"), "-o", str(out)], check=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_check_async_lds.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # the other thing the compiler cannot see inside inline asm: the stores with a scalar base (TH_STORE_U32_SADDR) must not read a base
    # register that a vector instruction (the v_readlane of a spilled pointer) wrote fewer than five wait states before -- round 4 traced
    # the GPU memory faults of two abandoned POA variants to exactly that (tools/isa_check_saddr_hazard.py)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_check_saddr_hazard.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_isa_checker_follows_control_flow(tmp_path):
    """the checker of the untracked LDS read on hand-written assembly: the case that got past its first version (round 5) -- the request sits at the bottom of the trip
    loop's body, the wait at the loop's top, and the register allocator has put a copy of the value IN FRONT of the wait: reached only through the back edge -- a read
    whose destination is used only after a wait on every path (clean), and a path that reaches a use through a conditional branch around the wait"""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "isa_check_async_lds.py")
    kernel = "_Z13th_k_chain_dpILb1ELi512EEvPKji9ThDevParaPKiS4_S4_PiS5_Py"
    wrap = lambda body: "%s:\n%s\n\ts_endpgm\n\t.amdhsa_kernel %s\n\t.end_amdhsa_kernel\n" % (kernel, body, kernel)
    hoisted = wrap("""\tds_read_b64 v[22:23], v14
.LBB0_1:
\ts_waitcnt vmcnt(0)
\tv_mov_b64_e32 v[24:25], v[22:23]
\ts_waitcnt lgkmcnt(0)
\tv_add_u32_e32 v1, v24, v25
\tds_read_b64 v[22:23], v14
\ts_cmp_eq_u64 s[4:5], 0
\ts_cbranch_scc1 .LBB0_1""")
    clean = wrap("""\tds_read_b64 v[22:23], v14
.LBB0_1:
\ts_waitcnt lgkmcnt(0)
\tv_mov_b64_e32 v[24:25], v[22:23]
\tds_read_b64 v[22:23], v14
\ts_cmp_eq_u64 s[4:5], 0
\ts_cbranch_scc1 .LBB0_1
\ts_waitcnt vmcnt(0) lgkmcnt(0)
\tv_mov_b32_e32 v22, 0""")
    around = wrap("""\tds_read_b64 v[22:23], v14
\ts_cbranch_scc1 .LBB0_2
\ts_waitcnt lgkmcnt(0)
.LBB0_2:
\tv_add_u32_e32 v1, v22, v23""")
    for name, text, want in (("hoisted", hoisted, 1), ("clean", clean, 0), ("around", around, 1)):
        f = tmp_path / (name + ".s")
        f.write_text(text)
        r = subprocess.run([sys.executable, tool, str(f)], capture_output=True, text=True)
        assert r.returncode == want, (name, r.stdout)
