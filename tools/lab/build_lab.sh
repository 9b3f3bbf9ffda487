#!/bin/bash
# lab build of the library (knobs read from the environment at every call, in-kernel stamps) + the lab harnesses that link it
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
HIPCC=/opt/rocm/bin/hipcc
make -C $R/qllm_amd/csrc -j8 variant NAME=lab DEFS=-DQLLM_LAB 2>&1 | grep -v "^/opt/rocm\|^make" || true
$HIPCC --offload-arch=gfx950 -O3 -I $R/include -o $R/tools/lab/g4lab $R/tools/lab/g4lab.cpp -L $R/tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
# decode: the step through the C ABI (release library / lab library) and the round-5 bisect
$HIPCC --offload-arch=gfx950 -O3 -I $R/include -o $R/tools/lab/cbench $R/tools/lab/cbench.cpp -L $R/qllm_amd -lqllm_mi355x -Wl,-rpath,'$ORIGIN/../../qllm_amd'
$HIPCC --offload-arch=gfx950 -O3 -I $R/include -o $R/tools/lab/cbench_lab $R/tools/lab/cbench.cpp -L $R/tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
# batch 2..32 per launch kind (BASELINE configs[3] with --cfg3): release library / lab library
$HIPCC --offload-arch=gfx950 -O3 -I $R/include -o $R/tools/lab/gbench $R/tools/lab/gbench.cpp -L $R/qllm_amd -lqllm_mi355x -Wl,-rpath,'$ORIGIN/../../qllm_amd'
$HIPCC --offload-arch=gfx950 -O3 -I $R/include -o $R/tools/lab/gbench_lab $R/tools/lab/gbench.cpp -L $R/tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I $R/include -I $R/qllm_amd/csrc -o $R/tools/lab/dbisect $R/tools/lab/dbisect.hip -L $R/tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
ls -la $R/tools/lab/g4lab $R/tools/lab/cbench $R/tools/lab/cbench_lab $R/tools/lab/dbisect $R/tools/lab/libqllm_lab.so
# round 6: the chained-links prototype and the batch-1 kernel's A/B variants (its own copy of the kernel header: strip1_lab_kernel.hpp)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I $R/include -I $R/qllm_amd/csrc -I $R/tools/lab -o $R/tools/lab/chainlab $R/tools/lab/chainlab.hip -L $R/tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
