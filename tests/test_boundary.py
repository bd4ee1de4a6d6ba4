"""The drop-in boundary without Python in the way: a plain C caller compiled against include/tidehunter_amd.h, and the struct
layouts against the reference's own header (field names, order and types of mini_tandem_para / tandem_seq_t / seq_t)."""
import os
import re
import subprocess

import pytest

import th_util as U

ROOT = U.ROOT
HDR = os.path.join(ROOT, "include", "tidehunter_amd.h")
REF = "/root/reference/src"


def _build_caller(tmp_path):
    exe = str(tmp_path / "caller")
    libdir = os.path.join(ROOT, "tidehunter_amd")
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_caller", "caller.c"), "-o", exe, "-L", libdir, "-ltidehunter_amd",
                    "-Wl,-rpath," + libdir], check=True)
    return exe


def _fields(text, struct_end):
    """declared field names of the struct whose closing line is '} <struct_end>;', in order"""
    body = text[:text.index("} " + struct_end + ";")]
    body = body[body.rindex("typedef struct"):]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        base = re.match(r"\s*((?:const\s+)?[A-Za-z_][A-Za-z_0-9]*(?:\s+[A-Za-z_][A-Za-z_0-9]*)*?)\s*([\*\s]*[A-Za-z_][A-Za-z_0-9]*)\s*$", first)
        assert base, decl
        typ = re.sub(r"^(th_|mini_tandem_)", "", base.group(1).strip())
        for item in [base.group(2)] + rest:
            item = item.strip()
            names.append((typ.replace("kstring_t", "K").replace("seq_t", "S"), item.count("*"), item.replace("*", "").strip()))
    return names


def test_c_caller_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    exe = _build_caller(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_c_caller_runs_on_gpu")
    r = subprocess.run([exe, os.path.join(U.DATA, "test_50x4.fa")], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr and r.stdout == ""


@pytest.mark.gpu
def test_c_caller_runs_on_gpu(tmp_path):
    exe = _build_caller(tmp_path)
    r = subprocess.run([exe, os.path.join(U.DATA, "test_50x4.fa")], capture_output=True, text=True, check=True)
    # README.md:222
    assert r.stdout == "test_50x4\trep0\t4.0\t300\t51\t250\t50\t100.0\t0\t59,109,159,208\tCGATCGATCGGCATGCATGCATGCTAGTCGATGCATCGGGATCAGCTAGT\n"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_struct_layouts_mirror_the_reference_header():
    mine = open(HDR).read()
    ref = open(os.path.join(REF, "tidehunter.h")).read()
    assert _fields(mine, "th_para_t") == _fields(ref, "mini_tandem_para")
    assert _fields(mine, "th_tandem_seq_t") == _fields(ref, "tandem_seq_t")
    assert _fields(mine, "th_seq_t") == _fields(open(os.path.join(REF, "seq.h")).read(), "seq_t")
    assert _fields(mine, "th_kstring_t") == _fields(open(os.path.join(REF, "kseq.h")).read(), "kstring_t")
