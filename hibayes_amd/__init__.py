"""hibayes_amd — MI355X-native engine for hibayes' individual-level per-SNP Gibbs sampler.

Public surface mirrors the reference's for this path (YinLiLin/hibayes v3.1.0):
    ibrm()        R/bayes.r:121          read_plink()  R/read_plink.r:24
    Bayes()       src/Bayes.cpp:60       cutwind_by_bp / cutwind_by_num  src/cutwind.cpp
All compute runs in libhibayes_gpu.so (hand-written gfx950 HIP kernels behind include/hibayes_gpu.h).
"""
from ._lib import HibayesError, lib, LIB_PATH
from .bayes import Bayes, ibrm
from .sbayes import SBayesD, sbrm
from .engine import Context
from .plink import read_plink, read_table, decode_bed, attach_bigmatrix, read_bigmatrix, write_bigmatrix
from .windows import cutwind_by_bp, cutwind_by_num

__all__ = ["Bayes", "ibrm", "read_plink", "read_table", "decode_bed", "attach_bigmatrix", "read_bigmatrix", "write_bigmatrix", "Context", "cutwind_by_bp",
           "cutwind_by_num", "SBayesD", "sbrm", "HibayesError", "lib", "LIB_PATH"]
__version__ = "0.1.0"
