#!/bin/bash
# The host side of libpfnl_hip under AddressSanitizer (an ASAN build of the C-ABI shim): builds
# pfnl_amd/lib/libpfnl_hip_asan.so (make asan) unless it is there, then runs the given pytest selection (default: the op hooks,
# the forward tests and the host tests that go through the C-ABI) with the sanitizer runtime preloaded.
#   tools/run_asan.sh --host                    tools/asan_host_path.py: the host-pointer forward over all kernel families, numpy only.
#                                               THE form that runs on a GPU box: torch's bundled HIP runtime aborts at initialisation
#                                               under a preloaded sanitizer runtime (ROCm 7.2 image, no /opt/rocm/lib/asan), so the
#                                               pytest form below works for the tests that do not touch torch.cuda (the CPU suite).
#   tools/run_asan.sh [pytest args...]          e.g.  tools/run_asan.sh tests/test_capi_symbols.py tests/test_host.py -q
# detect_leaks=0: the interpreter and the HIP runtime hold allocations for the life of the process; protect_shadow_gap=0: the ROCm
# runtime maps memory inside the shadow gap.
set -e
cd "$(dirname "$0")/.."
[ -f pfnl_amd/lib/libpfnl_hip_asan.so ] || make -C pfnl_amd/csrc asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export PFNL_HIP_LIB=$PWD/pfnl_amd/lib/libpfnl_hip_asan.so LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=1:halt_on_error=1
if [ "$1" = "--host" ]; then exec python tools/asan_host_path.py; fi
[ $# -eq 0 ] && set -- tests/test_capi_symbols.py tests/test_host.py -q
exec python -m pytest "$@"
