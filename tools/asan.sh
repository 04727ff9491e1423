#!/bin/bash
# CPU sanitizer pass over libumx's host logic.  Builds the whole library a second time
# with -fsanitize=address,undefined on the HOST side only (hipcc leaves gfx950 device code uninstrumented: GPU ASan needs xnack+,
# which this pool does not offer) into unmicst_amd/libumx_asan.so and runs the tests that drive the planner, the graph builder, the
# shard plan, the TIFF decoders and umx_describe* without a GPU against it.  Build container only -- never on the GPU box.
#   usage: bash tools/asan.sh [log file]        (exit status = pytest's; the log keeps sanitizer reports, if any)
set -u
cd "$(dirname "$0")/.."
LOG=${1:-/tmp/umx_asan.log}
ROCM=${ROCM_PATH:-/opt/rocm}
RT=$(ls $ROCM/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
OBJ=unmicst_amd/_obj_asan; mkdir -p $OBJ
SRCS="umx_kernels umx_conv_f16 umx_conv_first umx_graph umx_plan umx_engine umx_host umx_tiff umx_shard umx_train_kernels umx_train"
FLAGS="-O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libasan -Wno-unused-value -Wno-unused-result -Wno-option-ignored"
pids=()
for s in $SRCS; do
  if [ ! -f $OBJ/$s.o ] || [ unmicst_amd/csrc/$s.hip -nt $OBJ/$s.o ] || [ unmicst_amd/csrc/umx_kernels.h -nt $OBJ/$s.o ] || [ unmicst_amd/csrc/umx_internal.h -nt $OBJ/$s.o ] || [ include/umx.h -nt $OBJ/$s.o ]; then
    $ROCM/bin/hipcc $FLAGS -c unmicst_amd/csrc/$s.hip -o $OBJ/$s.o 2> $OBJ/$s.err &
    pids+=($!)
  fi
done
rc=0; for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait $p || rc=1; }; done
[ $rc -ne 0 ] && { cat $OBJ/*.err | grep -v "warning:" | head -40; echo "asan build failed"; exit 2; }
$ROCM/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt -fsanitize=address,undefined -shared-libasan -o unmicst_amd/libumx_asan.so $(for s in $SRCS; do echo $OBJ/$s.o; done) -ldl || exit 2
{
  echo "# tools/asan.sh at $(git rev-parse --short HEAD 2>/dev/null) on $(date -u +%FT%TZ): host-side ASan + UBSan build of libumx ($RT)"
  echo "# tests: test_abi test_host_logic test_meta_wiring test_tiffio_cpu test_driver_cpu test_sharding_cpu (no GPU in this container)"
} > $LOG
# leak detection off: CPython itself never frees its arenas; halt on the first real error
UMX_LIB=$PWD/unmicst_amd/libumx_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=66 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_abi.py tests/test_host_logic.py tests/test_meta_wiring.py tests/test_tiffio_cpu.py tests/test_driver_cpu.py tests/test_sharding_cpu.py -q -x -p no:cacheprovider >> $LOG 2>&1
rc=$?
echo "# pytest exit status $rc; sanitizer reports in this log: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' $LOG)" >> $LOG
tail -5 $LOG
exit $rc
