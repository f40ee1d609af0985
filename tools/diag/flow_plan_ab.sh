#!/bin/bash
# Round-3 LDS plan of the flow kernels: default build (in-place Taylor buffers, deltas in ACT, eps in registers: 148 VGPRs, three
# workgroups per CU at C2) against -DVMC_FLOW_EPS_REGS=0 (eps in an LDS line: 128 VGPRs, 80 units) and -DVMC_FLOW_MIN_BLOCKS=4 (eps in
# registers capped at 128 VGPRs with 52 B of scratch: four workgroups per CU at C2), and VMCPDE_FLOW_INPLACE=0 (two Taylor buffers).
#   python -m vmc_pde_amd.build --variant epslds -DVMC_FLOW_EPS_REGS=0; python -m vmc_pde_amd.build --variant occ4 -DVMC_FLOW_MIN_BLOCKS=4
cd "$(dirname "$0")/../.."
for wl in C2 C3 C4; do
  for v in "" epslds occ4; do
    lib=vmc_pde_amd/_lib/libvmcpde${v:+_$v}.so
    [ -f "$lib" ] || continue
    echo "== $wl ${v:-default}"
    VMCPDE_LIB=$PWD/$lib KEEP_ACT=0 python tools/probe_flow.py $wl 2>&1 | tail -2
  done
  echo "== $wl default, VMCPDE_FLOW_INPLACE=0"
  VMCPDE_FLOW_INPLACE=0 KEEP_ACT=0 python tools/probe_flow.py $wl 2>&1 | tail -2
done
