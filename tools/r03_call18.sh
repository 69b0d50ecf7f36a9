#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
KNOCK_STEPS=1000 KNOCK_MASKS="exit at S0 end=0x2000000;exit after S1 parts=0xe000000;exit after S1 parts, no noise=0xe000002;exit after S1 parts, no MFMA=0xe000010;exit after S1 parts, neither=0xe000012;exit after S1 end=0x3000000;exit after S1 end, no noise=0x3000002;exit after S2b=0x5000000;exit after S2b, no noise=0x5000002;exit after S2b, no S2b=0x5000020;exit after S3=0x7000000;exit after S3, no S3=0x7000040;exit after S5=0xa000000;exit after S5, no S5=0xa000200;exit after S6 parts=0xf000000;exit after S6 parts, no S6=0xf000400" python tools/knockout.py 0 C1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/knock_exit_C1.txt
