cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in base xfpk; do DEQSCI_HIP_LIB=build/w16v/lib_$v.so WHATIF_PASSES=60 timeout 300 python tools/w16_whatif_time.py 2>&1 | grep "^{"; done; done
DEQSCI_HIP_LIB=build/w16v/lib_xfpk.so timeout 300 python tools/w16_check.py check 2>&1 | grep -v amdgpu | tail -15
