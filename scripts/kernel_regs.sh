#!/bin/bash
# Development aid: register / spill report of every kernel in one .hip file of torchain_amd/csrc
#   scripts/kernel_regs.sh den_tied_kernel [extra -D flags]
cd "$(dirname "$0")/../torchain_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -w -I../../include --offload-arch=gfx950 -munsafe-fp-atomics "$@" --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/regs_$f.o 2>&1 |
  awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /SGPRs Spill:/ {ss=$5} /VGPRs Spill:/ {sp=$5} /ScratchSize/ {sc=$5} /LDS Size/ {printf "%s vgpr %s vspill %s sspill %s scratch %s\n", name, v, sp, ss, sc}' |
  while read n rest; do echo "$(echo $n | c++filt | sed 's/tc::(anonymous namespace):://; s/(tc::DenParams)//; s/void //') $rest"; done
