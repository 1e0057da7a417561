import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from flexam_amd import hip as H
import check_attn_fp8 as C
for lib in ("tree", "maskall"):
    H.load_library(H.LIB_PATH if lib == "tree" else "/root/repo/tools/probes/libflexam_var_maskall.so")
    print(lib)
    C.case(1, 1, 300); C.case(1, 1, 320); C.case(1, 1, 512); C.case(1, 1, 576)
