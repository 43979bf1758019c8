// The three C-ABI entry points of the rendezvous region that need no device, as gbp_capi.cpp defines them — the sanitizer
// harness links gbp_comm.cpp directly (gbp_capi.cpp itself needs a GPU build).
#include "../../include/gbp_mi355x.h"
#include "../../gbp_poplar_amd/csrc/gbp_comm.hpp"

#include <string>

extern "C" {
size_t gbp_comm_region_bytes(uint32_t n_cams, int world) { return gbp::comm_region_bytes(n_cams, world); }
int gbp_comm_region_init(void* region, size_t bytes, uint32_t n_cams, int world) {
  return gbp::comm_region_init(region, bytes, n_cams, world) == 0 ? GBP_OK : GBP_ERR_INVALID;
}
void gbp_comm_region_abort(void* region) { gbp::comm_region_abort(region); }
int gbp_comm_region_selftest(void* region, int rank, int world, int rounds) {
  std::string err;
  return gbp::comm_region_selftest(region, rank, world, rounds, err) == 0 ? GBP_OK : GBP_ERR_COMM;
}
}
