"""The C-ABI library loads and exports every symbol include/hibayes_gpu.h declares; the ctypes
mirror of its structs has the C layout. No compute calls: this runs without a GPU."""
import ctypes
import os
import re
import subprocess

import hibayes_amd as H
from hibayes_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "hibayes_gpu.h")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(hb_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_every_declared_symbol_is_exported():
    lib = H.lib()
    fns = declared_functions()
    assert len(fns) >= 40
    for f in fns:
        assert hasattr(lib, f), "missing export: " + f
    assert sorted(_lib.SYMBOLS) == fns, "hibayes_amd/_lib.py SYMBOLS out of sync with the header"


def test_version_and_error_channel():
    lib = H.lib()
    assert lib.hb_abi_version() == 6
    assert b"gfx950" in lib.hb_version()
    assert lib.hb_device_count() >= 0
    assert lib.hb_exchange_count(50000) == 50016


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hibayes_gpu.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(hb_bayes_args),sizeof(hb_bayes_out),sizeof(hb_ctx_params),sizeof(hb_sweep_in),sizeof(hb_sweep_out),'
                   'sizeof(hb_sweep_timing),sizeof(hb_run_info),offsetof(hb_bayes_args,seed),offsetof(hb_bayes_args,ctx),'
                   'offsetof(hb_bayes_out,alpha_sd),sizeof(hb_launch_stats),offsetof(hb_bayes_args,genotype_bits));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.BayesArgs), ctypes.sizeof(_lib.BayesOut), ctypes.sizeof(_lib.CtxParams),
            ctypes.sizeof(_lib.SweepIn), ctypes.sizeof(_lib.SweepOut), ctypes.sizeof(_lib.SweepTiming),
            ctypes.sizeof(_lib.RunInfo), _lib.BayesArgs.seed.offset, _lib.BayesArgs.ctx.offset,
            _lib.BayesOut.alpha_sd.offset, ctypes.sizeof(_lib.LaunchStats), _lib.BayesArgs.genotype_bits.offset]
    assert got == want


def test_warm_state_and_last_state_layouts_match_the_header_and_the_oracle(tmp_path):
    """hb_warm_state (ABI 6) in the product's header, its ctypes mirror, and the oracle's hbo_warm are one layout: the parity
    tests hand ONE state to both sides."""
    from oracle import oracle as O
    src = tmp_path / "w.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hibayes_gpu.h"\n#include "hb_oracle.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(hb_warm_state),sizeof(hbo_warm),offsetof(hb_warm_state,pi),offsetof(hbo_warm,pi),offsetof(hb_warm_state,vargL),offsetof(hbo_warm,vargL),'
                   'offsetof(hb_bayes_args,warm),offsetof(hb_bayes_out,last),offsetof(hb_run_info,lambda2));return 0;}\n')
    exe = tmp_path / "w"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    W = _lib.WarmState
    assert got == [ctypes.sizeof(W), ctypes.sizeof(O.Warm), W.pi.offset, O.Warm.pi.offset, W.vargL.offset, O.Warm.vargL.offset,
                   _lib.BayesArgs.warm.offset, _lib.BayesOut.last.offset, _lib.RunInfo.lambda2.offset]
    assert got[0] == got[1] and got[2] == got[3] and got[4] == got[5]


def test_header_is_plain_c_and_cites_the_reference():
    # compiles as C (no torch / C++ types in the signatures) and names the interface it replaces
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", HDR])
    text = open(HDR).read()
    assert "src/RcppExports.cpp:16-50" in text and "src/Bayes.cpp:60-88" in text
