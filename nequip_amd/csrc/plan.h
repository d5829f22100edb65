// Internal: host-side plan + device table image shared by the TP kernels.
// Public contract: include/nequip_amd.h (nqa_plan_create replaces TensorProductScatter.__init__,
// nequip/nn/_tp_scatter_base.py:10-33).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nequip_amd.h"

namespace nqa {

// One 'uvu' instruction (path).  `type` selects the unrolled Clebsch-Gordan code (NQA_TYPE_ID).
struct InstrDev {
  int32_t type, l1, l2, l3;
  int32_t mul;              // channels u (mul of in1 == mul of out slot)
  int32_t x_off, x_su, x_sm;  // in1 block: element (u, i) at x_off + u*x_su + i*x_sm
  int32_t y_off;            // in2 block offset (mul2 == 1)
  int32_t o_off, o_su, o_sm;  // out slot: element (u, k) at o_off + u*o_su + k*o_sm
  int32_t w_off;            // weight column offset
  int32_t shared_out;       // 1 -> several instructions write this slot (atomic accumulate)
  double coeff;             // path normalisation sqrt((2l3+1)/n_paths_into_slot) * sqrt(path_weight)
};
static_assert(sizeof(InstrDev) == 64, "InstrDev layout");

// Work decomposition of forward / edge-backward: (instruction, 64-channel chunk).
struct ChunkDev {
  int32_t instr;
  int32_t u0;
  int32_t ypart_off;  // column offset of this chunk's dY partials
  int32_t pad;
};

// One in1 irrep block (feature-gradient work decomposition).
struct BlkDev {
  int32_t l, mul;
  int32_t x_off, x_su, x_sm;
  int32_t instr_begin, instr_end;  // range in blk_instr[]
  int32_t pad;
};

struct XChunkDev {
  int32_t blk;
  int32_t u0;
};

struct ImageLayout {
  int64_t off_instr, off_chunks, off_blks, off_blk_instr, off_xchunks, off_ycol_ptr, off_ycol_idx;
  int64_t total_bytes;
};

}  // namespace nqa

namespace nqa {
struct SpecEntry;
}

struct nqa_plan {
  const nqa::SpecEntry* spec = nullptr;  // structure-specialised kernels (gen_spec.py), if prebuilt
  int32_t uniform_mul = 0;               // common multiplicity when spec != nullptr
  std::string structure_key;
  int32_t dim_in1 = 0, dim_in2 = 0, dim_out = 0, weight_numel = 0;
  int32_t ypart_width = 0;
  int32_t out_needs_zero = 0;
  int32_t any_shared_out = 0;
  std::vector<nqa::InstrDev> instr;
  std::vector<nqa::ChunkDev> chunks;
  std::vector<nqa::BlkDev> blks;
  std::vector<int32_t> blk_instr;
  std::vector<nqa::XChunkDev> xchunks;
  std::vector<int32_t> ycol_ptr;  // [dim_in2 + 1]
  std::vector<int32_t> ycol_idx;  // partial columns contributing to each dY component
  nqa::ImageLayout layout{};
};

namespace nqa {
void set_error(const std::string& msg);
}
