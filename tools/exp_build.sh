# Sourced by the experiment scripts.  An experiment library is built into the in-tree lib/ like the product one; what keeps it from
# passing for a product build is the fingerprint both libraries embed, which covers the compile flags (ace-compiler_amd/build.py
# source_fingerprint): a process loads the library only while it carries the SAME $ACEHIP_EXTRA_HIPCC_FLAGS, any other process
# (a later test run, the driver) sees a mismatch and rebuilds.  exp_build "<flags>" exports the flags for the runs that follow.
_exp_do_build() { python3 -c "import sys, ace_compiler_amd as A; sys.modules['ace_compiler_amd.build'].build_rt()" > /dev/null 2>&1; }
exp_build() { export ACEHIP_EXTRA_HIPCC_FLAGS="$1"; _exp_do_build; }
exp_restore() { unset ACEHIP_EXTRA_HIPCC_FLAGS; _exp_do_build; }
