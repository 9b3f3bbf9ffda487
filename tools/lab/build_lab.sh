#!/bin/bash
# lab build of the library (knobs read from the environment at every call, in-kernel stamps) + the prefill lab harness
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
make -C $R/qllm_amd/csrc -j8 variant NAME=lab DEFS=-DQLLM_LAB 2>&1 | grep -v "^/opt/rocm\|^make" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -I $R/include -o $R/tools/lab/g4lab $R/tools/lab/g4lab.cpp -L $R/tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
ls -la $R/tools/lab/g4lab $R/tools/lab/libqllm_lab.so
