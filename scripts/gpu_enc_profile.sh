#!/bin/bash
# Encoder diagnostics of one corpus class: phase table of the profile build (alone and loaded) and SQ instruction counters.
# usage: bash scripts/gpu_enc_profile.sh <tag> [class]
tag=${1:-base}; cls=${2:-6}
out=gpurun_out/r3; mkdir -p $out
export LZ4MI_LIB=$PWD/lz4_amd/liblz4mi_prof.so
NB=256 timeout -k 10 120 python scripts/gpu_phase_enc.py $cls > $out/phase_enc_256_$tag.log 2>&1 || { cat $out/phase_enc_256_$tag.log; exit 1; }
NB=4096 timeout -k 10 120 python scripts/gpu_phase_enc.py $cls > $out/phase_enc_4096_$tag.log 2>&1 || { cat $out/phase_enc_4096_$tag.log; exit 1; }
unset LZ4MI_LIB
bash scripts/pmc_sq.sh $cls 4096 > $out/pmc_sq_$tag.log 2>&1
grep -v amdgpu.ids $out/phase_enc_256_$tag.log $out/phase_enc_4096_$tag.log $out/pmc_sq_$tag.log
