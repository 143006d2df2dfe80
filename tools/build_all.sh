#!/bin/bash
# both libraries from the current sources (they embed the fingerprint of csrc/ + include/ and refuse to run against each other otherwise)
cd "$(dirname "$0")/.." && python3 -c "
import sys; sys.path.insert(0,'.')
import ace_compiler_amd as A
b=sys.modules['ace_compiler_amd.build']; print(A.build()); print(b.build_rt()); A.load_library(); print('both libraries built and loadable')" 2>&1 | grep -v "warning\|base_conv_batch16\|\^\|^ *83 "
