// The launchers of csrc/gbp_kernels.hip, for the CPU sanitizer build of the C-ABI (tests/test_host_sanitizers.py): the host side of the
// library — gbp_api_*.cpp — is compiled with g++ -fsanitize=address,undefined and linked against these stand-ins instead of the device
// code.  Nothing here computes: a launch without a device is a bug of the test, so every launcher aborts; only the pure host functions
// (grid sizing) return what the real ones return for an unlaunchable graph.
#include "../../gbp_poplar_amd/csrc/gbp_kernels.h"

#include <cstdio>
#include <cstdlib>

namespace gbp {

[[noreturn]] static void no_device(const char* what) {
  std::fprintf(stderr, "kernel_stubs: %s called in the CPU sanitizer build (no device code is linked)\n", what);
  std::abort();
}

void launch_sweep(const SweepArgs&, uint32_t, bool, hipStream_t, bool) { no_device("launch_sweep"); }
void launch_linearise(const SweepArgs&, uint32_t, hipStream_t) { no_device("launch_linearise"); }
void launch_beliefs(BeliefArgs, bool, bool, hipStream_t, bool) { no_device("launch_beliefs"); }
void launch_eval_fold(const EvalRide&, uint32_t, void*, hipStream_t) { no_device("launch_eval_fold"); }
void launch_eval_ride(const EvalRide&, const uint32_t*, const uint32_t*, const float4*, const float4*, const float*, hipStream_t) { no_device("launch_eval_ride"); }
uint32_t persist_blocks(uint32_t n_tiles, uint32_t, uint32_t, bool) { return (n_tiles + 3) / 4; }
int persist_max_resident_blocks() { return 0; }
PersistGrid persist_grid(uint32_t n_tiles, uint32_t, uint32_t, bool) { return PersistGrid{(n_tiles + 3) / 4, 0u, 0u}; }
hipError_t launch_persist(PersistArgs, bool, hipStream_t) { no_device("launch_persist"); }
void launch_copy_segments(const CopySegs&, const unsigned*, hipStream_t) { no_device("launch_copy_segments"); }
bool persist_probe(uint32_t, uint32_t, uint32_t, unsigned*, unsigned*, volatile unsigned*, bool, hipStream_t) { return false; }
void launch_state_get(const float4*, float*, int*, uint32_t, hipStream_t) { no_device("launch_state_get"); }
void launch_state_set(float4*, const int*, const uint32_t*, uint32_t, hipStream_t) { no_device("launch_state_set"); }
void launch_upload_scatter(float4*, float4*, const float4*, const float*, uint32_t, hipStream_t) { no_device("launch_upload_scatter"); }
void launch_means(const float4*, const float4*, float*, float*, uint32_t, uint32_t, unsigned long long*, unsigned long long*, bool, hipStream_t) { no_device("launch_means"); }
void launch_eval(const uint32_t*, const uint32_t*, const float4*, const float4*, const float*, const float*, const float*, int, DeviceEval*,
                 unsigned long long*, unsigned long long*, uint32_t, hipStream_t) { no_device("launch_eval"); }
uint32_t eval_blocks(uint32_t n_tiles) { return (n_tiles + 3) / 4; }
bool lab_launch_sweep_ablated(const SweepArgs&, uint32_t, int, hipStream_t) { return false; }
bool launch_flow_torture(float4*, unsigned long long*, int, int, int, unsigned, unsigned, int, hipStream_t) { no_device("launch_flow_torture"); }
bool debug_math_widths(int, int*, int*) { return false; }
void launch_debug_math(int, const float*, float*, int, hipStream_t) { no_device("launch_debug_math"); }

}  // namespace gbp
