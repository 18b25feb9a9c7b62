#!/bin/bash
# The library's HOST side (scheduler thread, job merging / splitting, pinned pool, result views) under AddressSanitizer on the GPU box:
# device code is not instrumented (-fno-gpu-sanitize; GPU ASan is not available on this pool).  Builds rafft_amd/libraffthip_asan.so
# when it is missing, then runs the asynchronous-API tests, the regrowth / split tests and the scheduler stress run against it.
#   tools/asan_host.sh        -> gpurun_out/asan_host.log (+ asan.<pid> reports if anything is found)
R=${GRAFT_REPO_ROOT:-$PWD}
LIB=$R/rafft_amd/libraffthip_asan.so
if [ ! -f $LIB ]; then
  (cd $R/rafft_amd/csrc && /opt/rocm/bin/hipcc -O3 -g --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -Wno-unused-function -Wno-missing-braces \
     -fsanitize=address -fno-gpu-sanitize -shared-libasan rafft_api.hip -o $LIB) || exit 1
fi
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p $R/gpurun_out
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:log_path=$R/gpurun_out/asan
export RAFFT_LIB=$LIB
cd $R
LD_PRELOAD=$RT timeout -k 10 500 python3 -m pytest tests/test_gpu_async.py "tests/test_gpu_parity.py::test_gpu_job_whose_candidate_table_would_outgrow_its_31_bit_slot_ids_is_split" \
   "tests/test_gpu_parity.py::test_gpu_more_productive_regions_than_the_short_lists_hold" -x -q -p no:cacheprovider > $R/gpurun_out/asan_host.log 2>&1
echo "pytest rc=$?" >> $R/gpurun_out/asan_host.log
LD_PRELOAD=$RT timeout -k 10 400 python3 tools/stress_scheduler.py >> $R/gpurun_out/asan_host.log 2>&1
echo "stress rc=$?" >> $R/gpurun_out/asan_host.log
ls $R/gpurun_out/asan.* 2>/dev/null | head
# optional second leg: the whole GPU suite against the instrumented library (ASAN_SUITE=1) - without the kinetics tests, whose torch.cuda
# initialisation does not survive the preloaded ASan runtime (dlopen of a torch library fails: not this library's code)
if [ -n "$ASAN_SUITE" ]; then
  LD_PRELOAD=$RT timeout -k 10 1000 python3 -m pytest tests -m gpu -q -p no:cacheprovider --ignore=tests/test_gpu_kinetics.py > $R/gpurun_out/asan_suite.log 2>&1
  echo "suite rc=$?" >> $R/gpurun_out/asan_suite.log
  ls $R/gpurun_out/asan.* 2>/dev/null | head
fi
