// update_gen.hip -- the update kernels (update_kernels.hpp) instantiated for GENERAL units: pooled VarLenSparseFeat fields
// (the reference's SequencePoolingLayer backward + embedding_dense_backward, inputs.py:141-155, sequence.py:49-77) and tables
// shared through `embedding_name` (inputs.py:158-180).  A translation unit of its own so that it compiles beside update.hip.
#include "update_kernels.hpp"

namespace dctr {
int launch_update_gen(const void* args, int vec, int lpr, int opt, unsigned grid_x, hipStream_t s) {
  const UpdArgs& a = *static_cast<const UpdArgs*>(args);
  const dim3 grid(grid_x), block(kThreads);
#define DCTR_UPD_GEN true
#include "update_launch.inc"
#undef DCTR_UPD_GEN
  return launch_status();
}
}  // namespace dctr
