// Plan construction for the fused tensor-product/scatter kernels (host only).
//
// Replaces the bookkeeping that e3nn's TensorProduct constructor does for the reference
// (nequip/nn/_tp_scatter_base.py:24-31; semantics restated in SURVEY.md Appendix A.2):
//   * mul_ir offsets of every irrep block,
//   * weight layout: instruction-list order, mul_in1*mul_in2 values each ('uvu', mul_in2 == 1),
//   * path coefficient sqrt(alpha), alpha = (2 l_out + 1) / #instructions sharing the output slot
//     (irrep_normalization="component", path_normalization="element", in/out variances 1),
//     times the instruction's path_weight.
#include <hip/hip_runtime.h>

#include "plan.h"

#include <cmath>
#include <cstring>
#include <sstream>

#include "generated/cg_generated.h"
#include "tp_spec.h"

namespace nqa {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

static int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

SpecEntry*& spec_registry_head() {
  static SpecEntry* head = nullptr;
  return head;
}

const SpecEntry* find_spec(const std::string& key) {
  for (const SpecEntry* e = spec_registry_head(); e != nullptr; e = e->next)
    if (e->key == key) return e;
  return nullptr;
}

}  // namespace nqa

using namespace nqa;

extern "C" {

int nqa_abi_version(void) { return NQA_ABI_VERSION; }
const char* nqa_last_error(void) { return g_last_error.c_str(); }
int nqa_lmax(void) { return NQA_LMAX; }

int nqa_plan_create(int32_t n_in1, const int32_t* in1_mul, const int32_t* in1_l, const int32_t* in1_p,
                    int32_t n_in2, const int32_t* in2_mul, const int32_t* in2_l, const int32_t* in2_p,
                    int32_t n_out, const int32_t* out_mul, const int32_t* out_l, const int32_t* out_p,
                    int32_t n_instr, const int32_t* instr_i1, const int32_t* instr_i2,
                    const int32_t* instr_io, const double* instr_path_weight, int32_t layout_in1,
                    int32_t layout_out, nqa_plan** plan_out) {
  if (plan_out == nullptr) {
    set_error("nqa_plan_create: plan pointer is NULL");
    return NQA_ERR_INVALID;
  }
  *plan_out = nullptr;
  auto fail = [&](int code, const std::string& m) {
    set_error("nqa_plan_create: " + m);
    return code;
  };
  if (n_in1 < 0 || n_in2 < 0 || n_out < 0 || n_instr < 0) return fail(NQA_ERR_INVALID, "negative count");
  if ((layout_in1 != NQA_LAYOUT_MUL_IR && layout_in1 != NQA_LAYOUT_IR_MUL) ||
      (layout_out != NQA_LAYOUT_MUL_IR && layout_out != NQA_LAYOUT_IR_MUL))
    return fail(NQA_ERR_INVALID, "unknown layout");

  auto check_irreps = [&](int n, const int32_t* mul, const int32_t* l, const int32_t* p, const char* name,
                          bool need_mul1) -> int {
    for (int i = 0; i < n; ++i) {
      if (mul[i] < 0 || l[i] < 0 || (p[i] != 1 && p[i] != -1)) {
        std::ostringstream os;
        os << name << "[" << i << "] invalid (mul=" << mul[i] << ", l=" << l[i] << ", p=" << p[i] << ")";
        return fail(NQA_ERR_INVALID, os.str());
      }
      if (l[i] > NQA_LMAX) {
        std::ostringstream os;
        os << name << "[" << i << "] has l=" << l[i] << " > supported l_max=" << NQA_LMAX;
        return fail(NQA_ERR_UNSUPPORTED, os.str());
      }
      if (need_mul1 && mul[i] != 1) {
        std::ostringstream os;
        os << name << "[" << i << "] has mul=" << mul[i] << "; 'uvu' with per-edge weights needs mul 1";
        return fail(NQA_ERR_UNSUPPORTED, os.str());
      }
    }
    return NQA_OK;
  };
  int rc;
  if ((rc = check_irreps(n_in1, in1_mul, in1_l, in1_p, "in1", false)) != NQA_OK) return rc;
  // only edge-attribute irreps that an instruction actually uses must have mul == 1 (checked below)
  if ((rc = check_irreps(n_in2, in2_mul, in2_l, in2_p, "in2", false)) != NQA_OK) return rc;
  if ((rc = check_irreps(n_out, out_mul, out_l, out_p, "out", false)) != NQA_OK) return rc;

  auto* P = new nqa_plan();
  std::vector<int32_t> off1(n_in1 + 1, 0), off2(n_in2 + 1, 0), offo(n_out + 1, 0);
  for (int i = 0; i < n_in1; ++i) off1[i + 1] = off1[i] + in1_mul[i] * (2 * in1_l[i] + 1);
  for (int i = 0; i < n_in2; ++i) off2[i + 1] = off2[i] + in2_mul[i] * (2 * in2_l[i] + 1);
  for (int i = 0; i < n_out; ++i) offo[i + 1] = offo[i] + out_mul[i] * (2 * out_l[i] + 1);
  P->dim_in1 = off1[n_in1];
  P->dim_in2 = off2[n_in2];
  P->dim_out = offo[n_out];

  std::vector<int> n_into(n_out, 0);
  for (int q = 0; q < n_instr; ++q) {
    const int i1 = instr_i1[q], i2 = instr_i2[q], io = instr_io[q];
    if (i1 < 0 || i1 >= n_in1 || i2 < 0 || i2 >= n_in2 || io < 0 || io >= n_out) {
      delete P;
      return fail(NQA_ERR_INVALID, "instruction index out of range");
    }
    n_into[io] += in2_mul[i2];  // 'uvu': num_elements = mul_in2
  }

  int32_t w_off = 0;
  for (int q = 0; q < n_instr; ++q) {
    const int i1 = instr_i1[q], i2 = instr_i2[q], io = instr_io[q];
    const int l1 = in1_l[i1], l2 = in2_l[i2], l3 = out_l[io];
    std::ostringstream os;
    os << "instruction " << q << " (" << i1 << "," << i2 << "," << io << "): ";
    if (in2_mul[i2] != 1) {
      delete P;
      return fail(NQA_ERR_UNSUPPORTED, os.str() + "edge-attribute multiplicity must be 1");
    }
    if (in1_mul[i1] != out_mul[io]) {
      delete P;
      return fail(NQA_ERR_INVALID, os.str() + "'uvu' requires mul_in1 == mul_out");
    }
    if (l3 < std::abs(l1 - l2) || l3 > l1 + l2) {
      delete P;
      return fail(NQA_ERR_INVALID, os.str() + "triangle rule |l1-l2| <= l3 <= l1+l2 violated");
    }
    if (in1_p[i1] * in2_p[i2] != out_p[io]) {
      delete P;
      return fail(NQA_ERR_INVALID, os.str() + "parity rule p1*p2 == p3 violated");
    }
    const double pw = instr_path_weight ? instr_path_weight[q] : 1.0;
    if (!(pw >= 0.0)) {
      delete P;
      return fail(NQA_ERR_INVALID, os.str() + "negative path weight");
    }
    InstrDev d{};
    d.type = NQA_TYPE_ID(l1, l2, l3);
    d.l1 = l1;
    d.l2 = l2;
    d.l3 = l3;
    d.mul = in1_mul[i1];
    d.x_off = off1[i1];
    if (layout_in1 == NQA_LAYOUT_MUL_IR) {
      d.x_su = 2 * l1 + 1;
      d.x_sm = 1;
    } else {
      d.x_su = 1;
      d.x_sm = in1_mul[i1];
    }
    d.y_off = off2[i2];
    d.o_off = offo[io];
    if (layout_out == NQA_LAYOUT_MUL_IR) {
      d.o_su = 2 * l3 + 1;
      d.o_sm = 1;
    } else {
      d.o_su = 1;
      d.o_sm = out_mul[io];
    }
    d.w_off = w_off;
    d.shared_out = n_into[io] > 1 ? 1 : 0;
    d.coeff = std::sqrt((double)(2 * l3 + 1) / (double)n_into[io] * pw);
    w_off += d.mul;  // mul_in1 * mul_in2
    if (d.shared_out) P->any_shared_out = 1;
    if (d.mul > 0) P->instr.push_back(d);
  }
  P->weight_numel = w_off;

  // output needs pre-zeroing if any slot is shared (atomic accumulation) or not written at all
  P->out_needs_zero = P->any_shared_out;
  for (int io = 0; io < n_out; ++io)
    if (n_into[io] == 0 && out_mul[io] > 0) P->out_needs_zero = 1;

  // forward / edge-backward chunks
  int32_t yp = 0;
  for (size_t q = 0; q < P->instr.size(); ++q) {
    const InstrDev& d = P->instr[q];
    for (int u0 = 0; u0 < d.mul; u0 += 64) {
      ChunkDev c{};
      c.instr = (int32_t)q;
      c.u0 = u0;
      c.ypart_off = yp;
      yp += 2 * d.l2 + 1;
      P->chunks.push_back(c);
    }
  }
  P->ypart_width = yp;

  // dY column map: for each component s of in2, the partial columns that must be summed
  P->ycol_ptr.assign(P->dim_in2 + 1, 0);
  std::vector<std::vector<int32_t>> cols(P->dim_in2);
  for (const ChunkDev& c : P->chunks) {
    const InstrDev& d = P->instr[c.instr];
    for (int j = 0; j < 2 * d.l2 + 1; ++j) cols[d.y_off + j].push_back(c.ypart_off + j);
  }
  for (int s = 0; s < P->dim_in2; ++s) {
    P->ycol_ptr[s + 1] = P->ycol_ptr[s] + (int32_t)cols[s].size();
    for (int32_t v : cols[s]) P->ycol_idx.push_back(v);
  }

  // in1 blocks and their instructions (feature-gradient decomposition)
  for (int i1 = 0; i1 < n_in1; ++i1) {
    if (in1_mul[i1] == 0) continue;
    BlkDev b{};
    b.l = in1_l[i1];
    b.mul = in1_mul[i1];
    b.x_off = off1[i1];
    if (layout_in1 == NQA_LAYOUT_MUL_IR) {
      b.x_su = 2 * b.l + 1;
      b.x_sm = 1;
    } else {
      b.x_su = 1;
      b.x_sm = b.mul;
    }
    b.instr_begin = (int32_t)P->blk_instr.size();
    for (size_t q = 0; q < P->instr.size(); ++q)
      if (P->instr[q].x_off == b.x_off) P->blk_instr.push_back((int32_t)q);
    b.instr_end = (int32_t)P->blk_instr.size();
    const int32_t bi = (int32_t)P->blks.size();
    P->blks.push_back(b);
    for (int u0 = 0; u0 < b.mul; u0 += 64) P->xchunks.push_back(XChunkDev{bi, u0});
  }

  // structure key (must match Structure.key() in gen_spec.py) and specialised-kernel lookup: needs a common
  // multiplicity over all feature / output irreps, mul_ir layouts and unit path weights
  {
    bool uniform = layout_in1 == NQA_LAYOUT_MUL_IR && layout_out == NQA_LAYOUT_MUL_IR && n_in1 > 0 && n_instr > 0;
    const int32_t m0 = n_in1 > 0 ? in1_mul[0] : 0;
    if (m0 <= 0) uniform = false;
    for (int i = 0; i < n_in1 && uniform; ++i) uniform = in1_mul[i] == m0;
    for (int i = 0; i < n_out && uniform; ++i) uniform = out_mul[i] == m0;
    for (int i = 0; i < n_in2 && uniform; ++i) uniform = in2_mul[i] == 1;
    for (int q = 0; q < n_instr && uniform; ++q) uniform = !instr_path_weight || instr_path_weight[q] == 1.0;
    std::ostringstream key;
    key << "i1:";
    for (int i = 0; i < n_in1; ++i) key << (i ? "," : "") << in1_l[i];
    key << "|i2:";
    for (int i = 0; i < n_in2; ++i) key << (i ? "," : "") << in2_l[i];
    key << "|o:";
    for (int i = 0; i < n_out; ++i) key << (i ? "," : "") << out_l[i];
    key << "|p:";
    for (int q = 0; q < n_instr; ++q) key << (q ? "," : "") << instr_i1[q] << "-" << instr_i2[q] << "-" << instr_io[q];
    P->structure_key = key.str();
    if (uniform) {
      P->spec = find_spec(P->structure_key);
      P->uniform_mul = m0;
    }
  }

  // device image layout
  ImageLayout& L = P->layout;
  int64_t o = 0;
  L.off_instr = o;
  o = align_up(o + (int64_t)P->instr.size() * sizeof(InstrDev), 64);
  L.off_chunks = o;
  o = align_up(o + (int64_t)P->chunks.size() * sizeof(ChunkDev), 64);
  L.off_blks = o;
  o = align_up(o + (int64_t)P->blks.size() * sizeof(BlkDev), 64);
  L.off_blk_instr = o;
  o = align_up(o + (int64_t)P->blk_instr.size() * sizeof(int32_t), 64);
  L.off_xchunks = o;
  o = align_up(o + (int64_t)P->xchunks.size() * sizeof(XChunkDev), 64);
  L.off_ycol_ptr = o;
  o = align_up(o + (int64_t)P->ycol_ptr.size() * sizeof(int32_t), 64);
  L.off_ycol_idx = o;
  o = align_up(o + (int64_t)P->ycol_idx.size() * sizeof(int32_t), 64);
  L.total_bytes = o > 0 ? o : 64;

  *plan_out = P;
  return NQA_OK;
}

void nqa_plan_destroy(nqa_plan* plan) { delete plan; }

int64_t nqa_plan_query(const nqa_plan* plan, int32_t field) {
  if (plan == nullptr) return -1;
  switch (field) {
    case NQA_PLAN_DIM_IN1: return plan->dim_in1;
    case NQA_PLAN_DIM_IN2: return plan->dim_in2;
    case NQA_PLAN_DIM_OUT: return plan->dim_out;
    case NQA_PLAN_WEIGHT_NUMEL: return plan->weight_numel;
    case NQA_PLAN_NUM_INSTR: return (int64_t)plan->instr.size();
    case NQA_PLAN_OUT_NEEDS_ZERO: return plan->out_needs_zero;
    case NQA_PLAN_YPART_WIDTH: return plan->ypart_width;
    case NQA_PLAN_HAS_SPECIALIZED: return plan->spec != nullptr ? 1 : 0;
    case NQA_PLAN_FUSED_ROWS_OK:
      // measured on the cfg-3 box at mul = 32: 183 values (l_max 4, inputs l <= 2) 2.6 ms fused vs 3.1 ms in two
      // kernels; 364 values (full l_max 4 middle layer, 338-539 spilled registers) 74.6 ms vs 8.7 ms
      return plan->spec != nullptr && plan->spec->od + 2 * plan->spec->xd + 2 * plan->spec->np <= 200 ? 1 : 0;
    default: return -1;
  }
}

int64_t nqa_plan_image_bytes(const nqa_plan* plan) { return plan ? plan->layout.total_bytes : -1; }

int nqa_plan_image_write(const nqa_plan* plan, void* host_dst, int64_t host_dst_bytes) {
  if (plan == nullptr || host_dst == nullptr) {
    set_error("nqa_plan_image_write: NULL argument");
    return NQA_ERR_INVALID;
  }
  const ImageLayout& L = plan->layout;
  if (host_dst_bytes < L.total_bytes) {
    set_error("nqa_plan_image_write: destination too small");
    return NQA_ERR_WORKSPACE;
  }
  char* d = static_cast<char*>(host_dst);
  std::memset(d, 0, (size_t)L.total_bytes);
  auto put = [&](int64_t off, const void* src, size_t bytes) {
    if (bytes) std::memcpy(d + off, src, bytes);
  };
  put(L.off_instr, plan->instr.data(), plan->instr.size() * sizeof(InstrDev));
  put(L.off_chunks, plan->chunks.data(), plan->chunks.size() * sizeof(ChunkDev));
  put(L.off_blks, plan->blks.data(), plan->blks.size() * sizeof(BlkDev));
  put(L.off_blk_instr, plan->blk_instr.data(), plan->blk_instr.size() * sizeof(int32_t));
  put(L.off_xchunks, plan->xchunks.data(), plan->xchunks.size() * sizeof(XChunkDev));
  put(L.off_ycol_ptr, plan->ycol_ptr.data(), plan->ycol_ptr.size() * sizeof(int32_t));
  put(L.off_ycol_idx, plan->ycol_idx.data(), plan->ycol_idx.size() * sizeof(int32_t));
  return NQA_OK;
}

}  // extern "C"
