#!/bin/bash
# Timing-only builds of k_numeric on the GPU box (ablations / tuning macros), 148^3 assembly.  The ablation switches live
# in the LAB build (stan_amd/csrc/lab/lab_hooks.patch applied to copies of the sources: `make lab`), not in the product.
# usage: bash tools/asm_lab.sh "-DSTAN_ABL=1" "-DSTAN_ABL=2" ...
#   STAN_ABL: 1 no block arithmetic (phase B), 2 no ordered accumulation (phase D), 3 no write-out, 4 no slot search,
#             5 no Jacobians (phase A), 6 no incidence chunks at all (set-up, chains, write-out only)
exec bash $(dirname $0)/lib_lab.sh tools/asm_time.py "$@"
