#!/bin/bash
# builds the variants of the 8-phase GEMM probe as separate binaries (co-compiled variants perturb each other's code generation)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value"
build() { name=$1; shift; $HIPCC $FLAGS "$@" gemm_8phase.hip -o bin/gemm_8phase_$name & }
build base
build nostagger -DNO_STAGGER
build nosetprio -DNO_SETPRIO
build st16x32 -DLAYOUT=0
build mfma32 -DMFMA32
build bf16 -DBF16
wait
ls -la bin/gemm_8phase_*
