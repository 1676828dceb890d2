"""examples/ibrm_demo.c compiles and links against the header and the library without a GPU (it then fails loudly at
hb_ctx_create: there is no CPU fallback); shim/Bayes_gpu.cpp names every argument of the reference's Bayes()."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_example_compiles_links_and_refuses_to_run_without_a_device(tmp_path):
    import hibayes_amd as H
    exe = str(tmp_path / "ibrm_demo")
    libdir = os.path.join(ROOT, "hibayes_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "ibrm_demo.c"), "-L", libdir, "-lhibayes_gpu", "-Wl,-rpath," + libdir, "-o", exe])
    if H.lib().hb_device_count() == 0:
        p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "demo", "demo")], capture_output=True)
        assert p.returncode == 1 and b"no HIP device available" in p.stderr


def test_shim_forwards_all_27_arguments_of_the_reference_signature():
    src = open(os.path.join(ROOT, "shim", "Bayes_gpu.cpp")).read()
    sig = src[src.index("Rcpp::List Bayes("):src.index("{", src.index("Rcpp::List Bayes("))]
    names = re.findall(r"(\w+)\s*(?:=\s*[\w_]+)?\s*[,)]", sig)
    want = ["y", "X", "model", "Pi", "Kival", "Ki", "C", "R", "fold", "niter", "nburn", "thin", "epsl_y_J", "epsl_Gi", "epsl_index",
            "dfvr", "s2vr", "vg", "dfvg", "s2vg", "ve", "dfve", "s2ve", "windindx", "outfreq", "threads", "verbose"]   # src/Bayes.cpp:60-88
    assert [n for n in names if n in want] == want
    body = src[src.index("hb_bayes_args a = {};"):]
    for n in want:
        assert re.search(r"\b%s\b" % n, body), "argument %s is not forwarded" % n
    assert "levels_of(" in src and "static CharacterVector levels_of" in src
