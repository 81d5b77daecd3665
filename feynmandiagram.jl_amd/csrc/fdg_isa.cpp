// gfx950 assembly printer for the optimized program (fdg_opt.h): one wave of 64
// lanes owns 64 samples (lane = sample) and runs the graph as straight-line
// code -- every micro-op is exactly one VALU instruction:
//     M_MUL  -> v_mul_f64      M_ADD -> v_add_f64      M_MULC -> v_mul_f64 (const)
// (neg source modifiers carry factor -1), with values in VGPR pairs, the first
// overflow level in LDS columns lds[slot][lane] (ds_read/write_b64, conflict
// free: consecutive lanes hit consecutive 8-byte words), the second in the
// wave's HBM panel ws[slot][lane] (global_load/store_dwordx2, one coalesced
// 512-byte transaction per access).  The instruction stream *is* the node table.
//
// A persistent grid is used: each wave loops over tiles of 64 samples.  No
// barriers, no cross-wave traffic.  s_waitcnt is placed by an exact model of the
// in-order vmcnt / lgkmcnt counters.
//
// Kernel arguments (all 8 bytes): leaf, ss, ls, root, rs, rk, ws, B, nwg
//   leaf value i of sample b: leaf[b*ss + i*ls]; root k: root[b*rs + k*rk]
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "fdg_opt.h"

namespace fdg {
namespace {

constexpr int V_LANE8 = 1, V_LEAFOFF = 2, V_ROOTOFF = 3, V_TMP = 4, V_BASE = 6;
constexpr int S_LEAF = 4, S_SS = 6, S_LS = 8, S_ROOT = 10, S_RS = 12, S_RK = 14, S_WS = 16, S_B = 18, S_NWG = 20;
constexpr int S_TILE = 22, S_NTILES = 23, S_LT = 24, S_RT = 26, S_PANEL = 28, S_LS8 = 30, S_RK8 = 32, S_A = 34,
              S_T = 36, S_C = 38, S_X = 40;  // S_X.. : scratch (4)
constexpr int S_END = 48;

struct Emit {
  std::ostringstream os;
  uint64_t vm_issued = 0, lg_issued = 0, vm_done = 0, lg_done = 0;
  // pending[reg] = (kind 0 none / 1 vm / 2 lgkm, seq)
  std::vector<std::pair<uint8_t, uint64_t>> pend;

  void ins(const std::string &s) { os << "\t" << s << "\n"; }
  std::string vr(uint32_t r) const {
    char b[32];
    std::snprintf(b, sizeof b, "v[%u:%u]", V_BASE + 2 * r, V_BASE + 2 * r + 1);
    return b;
  }
  void wait_reg(uint32_t r) {
    auto &p = pend[r];
    if (p.first == 1 && p.second > vm_done) {
      uint64_t n = vm_issued - p.second;            // ops issued after it may stay outstanding
      if (n > 63) n = 63;
      ins("s_waitcnt vmcnt(" + std::to_string(n) + ")");
      vm_done = std::max(vm_done, vm_issued - n);   // everything up to that seq has returned
      if (vm_done < p.second) vm_done = p.second;
    } else if (p.first == 2 && p.second > lg_done) {
      uint64_t n = lg_issued - p.second;
      if (n > 15) n = 15;
      ins("s_waitcnt lgkmcnt(" + std::to_string(n) + ")");
      lg_done = std::max(lg_done, lg_issued - n);
      if (lg_done < p.second) lg_done = p.second;
    }
    p.first = 0;
  }
  void drain() {
    ins("s_waitcnt vmcnt(0) lgkmcnt(0)");
    vm_done = vm_issued;
    lg_done = lg_issued;
    for (auto &p : pend) p.first = 0;
  }
};

std::string f64_inline(double f, bool &ok) {
  ok = true;
  if (f == 0.5) return "0.5";
  if (f == -0.5) return "-0.5";
  if (f == 2.0) return "2.0";
  if (f == -2.0) return "-2.0";
  if (f == 4.0) return "4.0";
  if (f == -4.0) return "-4.0";
  if (f == 1.0) return "1.0";
  if (f == -1.0) return "-1.0";
  ok = false;
  return "";
}

std::string hex32(uint32_t v) {
  char b[16];
  std::snprintf(b, sizeof b, "0x%08x", v);
  return b;
}

// s[dst:dst+1] = s[base:base+1] + k * s[mul8:mul8+1]   (k compile-time, 64-bit)
void emit_scaled_addr(Emit &E, int dst, int base, int mul8, uint32_t k) {
  if (k == 0) {
    E.ins("s_mov_b64 s[" + std::to_string(dst) + ":" + std::to_string(dst + 1) + "], s[" + std::to_string(base) + ":" +
          std::to_string(base + 1) + "]");
    return;
  }
  const std::string ks = hex32(k);
  E.ins("s_mul_i32 s" + std::to_string(S_X) + ", s" + std::to_string(mul8) + ", " + ks);
  E.ins("s_mul_hi_u32 s" + std::to_string(S_X + 1) + ", s" + std::to_string(mul8) + ", " + ks);
  E.ins("s_mul_i32 s" + std::to_string(S_X + 2) + ", s" + std::to_string(mul8 + 1) + ", " + ks);
  E.ins("s_add_u32 s" + std::to_string(S_X + 1) + ", s" + std::to_string(S_X + 1) + ", s" + std::to_string(S_X + 2));
  E.ins("s_add_u32 s" + std::to_string(dst) + ", s" + std::to_string(base) + ", s" + std::to_string(S_X));
  E.ins("s_addc_u32 s" + std::to_string(dst + 1) + ", s" + std::to_string(base + 1) + ", s" + std::to_string(S_X + 1));
}

// panel slot address: returns "s[a:b] offset:imm" operand text after emitting the SALU that forms the base
std::string panel_operand(Emit &E, uint32_t slot) {
  const uint64_t byte = (uint64_t)slot * 512u;
  const uint64_t hi = byte & ~4095ull, lo = byte & 4095ull;
  if (hi == 0) return "s[" + std::to_string(S_PANEL) + ":" + std::to_string(S_PANEL + 1) + "] offset:" + std::to_string(lo);
  E.ins("s_add_u32 s" + std::to_string(S_A) + ", s" + std::to_string(S_PANEL) + ", " + hex32((uint32_t)hi));
  E.ins("s_addc_u32 s" + std::to_string(S_A + 1) + ", s" + std::to_string(S_PANEL + 1) + ", " + hex32((uint32_t)(hi >> 32)));
  return "s[" + std::to_string(S_A) + ":" + std::to_string(S_A + 1) + "] offset:" + std::to_string(lo);
}

}  // namespace

std::string emit_isa(const Lowered &p, const OptProgram &prog, const std::string &kname) {
  Emit E;
  E.pend.assign(std::max<uint32_t>(prog.n_reg_used, 1), {0, 0});
  const uint32_t lds_bytes = prog.n_lds_used * 512u;
  const uint32_t panel_bytes_per_wave = std::max<uint32_t>(prog.n_mem_used, 1) * 512u;
  std::ostringstream &os = E.os;
  os << "\t.amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"\n\t.amdhsa_code_object_version 6\n\t.text\n";
  os << "\t.protected\t" << kname << "\n\t.globl\t" << kname << "\n\t.p2align\t8\n\t.type\t" << kname << ",@function\n";
  os << kname << ":\n";
  auto S = [](int r) { return "s" + std::to_string(r); };
  auto S2 = [](int r) { return "s[" + std::to_string(r) + ":" + std::to_string(r + 1) + "]"; };
  auto V = [](int r) { return "v" + std::to_string(r); };

  const char *dbg = std::getenv("FDG_ISA_DEBUG");
  const bool dbg_noleaf = dbg && std::strstr(dbg, "noleaf");
  const bool dbg_nolds = dbg && std::strstr(dbg, "nolds");
  const bool dbg_panelin = dbg && std::strstr(dbg, "panelin");   // timing only: read leaves as [tile][L][64]
  // ---- prologue ------------------------------------------------------------
  E.ins("s_load_dwordx8 s[4:11], s[0:1], 0x0");
  E.ins("s_load_dwordx8 s[12:19], s[0:1], 0x20");
  E.ins("s_load_dwordx2 s[20:21], s[0:1], 0x40");
  for (uint32_t r = 0; r < E.pend.size(); ++r) E.wait_reg(r);   // loads never consumed (evicted prefetches): no WAW into the next tile
  E.ins("s_waitcnt lgkmcnt(0)");
  E.ins("v_lshlrev_b32_e32 " + V(V_LANE8) + ", 3, v0");
  E.ins("v_mul_lo_u32 " + V(V_LEAFOFF) + ", v0, " + S(S_SS));          // lane*ss (low 32 bits)
  E.ins("v_lshlrev_b32_e32 " + V(V_LEAFOFF) + ", 3, " + V(V_LEAFOFF));
  E.ins("v_mul_lo_u32 " + V(V_ROOTOFF) + ", v0, " + S(S_RS));
  E.ins("v_lshlrev_b32_e32 " + V(V_ROOTOFF) + ", 3, " + V(V_ROOTOFF));
  E.ins("s_lshl_b64 " + S2(S_LS8) + ", " + S2(S_LS) + ", 3");
  E.ins("s_lshl_b64 " + S2(S_RK8) + ", " + S2(S_RK) + ", 3");
  // ntiles = (B + 63) >> 6 (B < 2^37 so that ntiles fits 32 bits)
  E.ins("s_add_u32 " + S(S_X) + ", " + S(S_B) + ", 63");
  E.ins("s_addc_u32 " + S(S_X + 1) + ", " + S(S_B + 1) + ", 0");
  E.ins("s_lshr_b64 " + S2(S_X) + ", " + S2(S_X) + ", 6");
  E.ins("s_mov_b32 " + S(S_NTILES) + ", " + S(S_X));
  E.ins("s_mov_b32 " + S(S_TILE) + ", s2");
  // panel base = ws + wg * panel_bytes_per_wave
  E.ins("s_mul_i32 " + S(S_X) + ", s2, " + hex32(panel_bytes_per_wave));
  E.ins("s_mul_hi_u32 " + S(S_X + 1) + ", s2, " + hex32(panel_bytes_per_wave));
  E.ins("s_add_u32 " + S(S_PANEL) + ", " + S(S_WS) + ", " + S(S_X));
  E.ins("s_addc_u32 " + S(S_PANEL + 1) + ", " + S(S_WS + 1) + ", " + S(S_X + 1));
  E.ins("s_cmp_ge_u32 " + S(S_TILE) + ", " + S(S_NTILES));
  E.ins("s_cbranch_scc0 .Ltile");
  E.ins("s_endpgm");
  os << ".Ltile:\n";
  // b0 = tile*64 (64-bit in S_X:S_X+1); valid lanes; exec
  E.ins("s_mov_b32 " + S(S_X + 1) + ", 0");
  E.ins("s_mov_b32 " + S(S_X) + ", " + S(S_TILE));
  E.ins("s_lshl_b64 " + S2(S_X) + ", " + S2(S_X) + ", 6");
  E.ins("s_sub_u32 " + S(S_T) + ", " + S(S_B) + ", " + S(S_X));           // B - b0 (low); high decides >= 64
  E.ins("s_subb_u32 " + S(S_T + 1) + ", " + S(S_B + 1) + ", " + S(S_X + 1));
  E.ins("s_cmp_lg_u32 " + S(S_T + 1) + ", 0");
  E.ins("s_cselect_b32 " + S(S_T) + ", 64, " + S(S_T));
  E.ins("s_min_u32 " + S(S_T) + ", " + S(S_T) + ", 64");
  E.ins("s_bfm_b64 " + S2(S_A) + ", " + S(S_T) + ", 0");
  E.ins("s_cmp_ge_u32 " + S(S_T) + ", 64");
  E.ins("s_cselect_b64 exec, -1, " + S2(S_A));
  // leaf tile base = leaf + b0*ss*8 ; root tile base = root + b0*rs*8   (b0 < 2^32 assumed hi part folded)
  auto tile_base = [&](int dst, int base, int stride) {
    E.ins("s_mul_i32 " + S(S_A) + ", " + S(S_X) + ", " + S(stride));
    E.ins("s_mul_hi_u32 " + S(S_A + 1) + ", " + S(S_X) + ", " + S(stride));
    E.ins("s_mul_i32 " + S(S_T) + ", " + S(S_X) + ", " + S(stride + 1));
    E.ins("s_add_u32 " + S(S_A + 1) + ", " + S(S_A + 1) + ", " + S(S_T));
    E.ins("s_mul_i32 " + S(S_T) + ", " + S(S_X + 1) + ", " + S(stride));
    E.ins("s_add_u32 " + S(S_A + 1) + ", " + S(S_A + 1) + ", " + S(S_T));
    E.ins("s_lshl_b64 " + S2(S_A) + ", " + S2(S_A) + ", 3");
    E.ins("s_add_u32 " + S(dst) + ", " + S(base) + ", " + S(S_A));
    E.ins("s_addc_u32 " + S(dst + 1) + ", " + S(base + 1) + ", " + S(S_A + 1));
  };
  tile_base(S_LT, S_LEAF, S_SS);
  tile_base(S_RT, S_ROOT, S_RS);
  if (dbg_panelin) {
    E.ins("s_mul_i32 " + S(S_A) + ", " + S(S_TILE) + ", " + hex32(p.L * 512u));
    E.ins("s_mul_hi_u32 " + S(S_A + 1) + ", " + S(S_TILE) + ", " + hex32(p.L * 512u));
    E.ins("s_add_u32 " + S(S_LT) + ", " + S(S_LEAF) + ", " + S(S_A));
    E.ins("s_addc_u32 " + S(S_LT + 1) + ", " + S(S_LEAF + 1) + ", " + S(S_A + 1));
  }

  // ---- body ------------------------------------------------------------------
  for (const MOp &o : prog.ops) {
    switch (o.kind) {
      case M_LD_LEAF: {
        if (dbg_noleaf) break;   // timing experiments only (results are garbage)
        E.wait_reg(o.d);
        if (dbg_panelin) {
          const uint64_t byte = (uint64_t)o.a * 512u;
          E.ins("s_add_u32 " + S(S_A) + ", " + S(S_LT) + ", " + hex32((uint32_t)(byte & ~4095ull)));
          E.ins("s_addc_u32 " + S(S_A + 1) + ", " + S(S_LT + 1) + ", 0");
          E.ins("global_load_dwordx2 " + E.vr(o.d) + ", " + V(V_LANE8) + ", " + S2(S_A) + " offset:" + std::to_string(byte & 4095ull));
          E.pend[o.d] = {1, ++E.vm_issued};
          break;
        }
        emit_scaled_addr(E, S_A, S_LT, S_LS8, o.a);
        E.ins("global_load_dwordx2 " + E.vr(o.d) + ", " + V(V_LEAFOFF) + ", " + S2(S_A));
        E.pend[o.d] = {1, ++E.vm_issued};
        break;
      }
      case M_LD_MEM: {
        E.wait_reg(o.d);
        const std::string opnd = panel_operand(E, o.a);
        E.ins("global_load_dwordx2 " + E.vr(o.d) + ", " + V(V_LANE8) + ", " + opnd);
        E.pend[o.d] = {1, ++E.vm_issued};
        break;
      }
      case M_ST_MEM: {
        E.wait_reg(o.a);
        const std::string opnd = panel_operand(E, o.d);
        E.ins("global_store_dwordx2 " + V(V_LANE8) + ", " + E.vr(o.a) + ", " + opnd);
        ++E.vm_issued;
        break;
      }
      case M_LD_LDS:
        if (dbg_nolds) break;
        E.wait_reg(o.d);
        E.ins("ds_read_b64 " + E.vr(o.d) + ", " + V(V_LANE8) + " offset:" + std::to_string(o.a * 512u));
        E.pend[o.d] = {2, ++E.lg_issued};
        break;
      case M_ST_LDS:
        if (dbg_nolds) break;
        E.wait_reg(o.a);
        E.ins("ds_write_b64 " + V(V_LANE8) + ", " + E.vr(o.a) + " offset:" + std::to_string(o.d * 512u));
        ++E.lg_issued;
        break;
      case M_MUL:
      case M_ADD: {
        E.wait_reg(o.a);
        E.wait_reg(o.b);
        E.wait_reg(o.d);
        E.ins(std::string(o.kind == M_MUL ? "v_mul_f64 " : "v_add_f64 ") + E.vr(o.d) + ", " + (o.nega ? "-" : "") +
              E.vr(o.a) + ", " + (o.negb ? "-" : "") + E.vr(o.b));
        break;
      }
      case M_MULC: {
        E.wait_reg(o.a);
        E.wait_reg(o.d);
        bool inl;
        std::string c = f64_inline(o.imm, inl);
        if (!inl) {
          uint64_t u;
          std::memcpy(&u, &o.imm, 8);
          E.ins("s_mov_b32 " + S(S_C) + ", " + hex32((uint32_t)u));
          E.ins("s_mov_b32 " + S(S_C + 1) + ", " + hex32((uint32_t)(u >> 32)));
          c = S2(S_C);
        }
        E.ins("v_mul_f64 " + E.vr(o.d) + ", " + (o.nega ? "-" : "") + E.vr(o.a) + ", " + c);
        break;
      }
      case M_LD_ACC:
        E.wait_reg(o.d);
        E.ins("v_accvgpr_read_b32 v" + std::to_string(V_BASE + 2 * o.d) + ", a" + std::to_string(2 * o.a));
        E.ins("v_accvgpr_read_b32 v" + std::to_string(V_BASE + 2 * o.d + 1) + ", a" + std::to_string(2 * o.a + 1));
        break;
      case M_ST_ACC:
        E.wait_reg(o.a);
        E.ins("v_accvgpr_write_b32 a" + std::to_string(2 * o.d) + ", v" + std::to_string(V_BASE + 2 * o.a));
        E.ins("v_accvgpr_write_b32 a" + std::to_string(2 * o.d + 1) + ", v" + std::to_string(V_BASE + 2 * o.a + 1));
        break;
      case M_ROOT: {
        E.wait_reg(o.a);
        emit_scaled_addr(E, S_A, S_RT, S_RK8, o.d);
        if (o.nega) {
          E.ins("v_mov_b32_e32 " + V(V_TMP) + ", v" + std::to_string(V_BASE + 2 * o.a));
          E.ins("v_xor_b32_e32 " + V(V_TMP + 1) + ", 0x80000000, v" + std::to_string(V_BASE + 2 * o.a + 1));
          E.ins("global_store_dwordx2 " + V(V_ROOTOFF) + ", v[" + std::to_string(V_TMP) + ":" + std::to_string(V_TMP + 1) +
                "], " + S2(S_A));
        } else {
          E.ins("global_store_dwordx2 " + V(V_ROOTOFF) + ", " + E.vr(o.a) + ", " + S2(S_A));
        }
        ++E.vm_issued;
        break;
      }
      default:
        break;
    }
  }
  // ---- next tile ---------------------------------------------------------------
  // No drain: every load whose value was read has been waited for, and the counters complete in
  // order, so a leftover store only makes the next iteration's waits conservative, never wrong.
  // (LDS / panel slots are re-written next tile; same-wave memory ops stay in program order.)
  for (uint32_t r = 0; r < E.pend.size(); ++r) E.wait_reg(r);   // loads never consumed (evicted prefetches): no WAW into the next tile
  E.ins("s_waitcnt lgkmcnt(0)");
  E.ins("s_add_u32 " + S(S_TILE) + ", " + S(S_TILE) + ", " + S(S_NWG));
  E.ins("s_cmp_ge_u32 " + S(S_TILE) + ", " + S(S_NTILES));
  E.ins("s_cbranch_scc0 .Lback");
  E.ins("s_endpgm");
  // long backward jump (the body may exceed the 16-bit branch range)
  os << ".Lback:\n";
  E.ins("s_getpc_b64 " + S2(S_A));
  os << ".Lpc:\n";
  E.ins("s_add_u32 " + S(S_A) + ", " + S(S_A) + ", (.Ltile-.Lpc)&0xffffffff");
  E.ins("s_addc_u32 " + S(S_A + 1) + ", " + S(S_A + 1) + ", (.Ltile-.Lpc)>>32");
  E.ins("s_setpc_b64 " + S2(S_A));
  E.ins("s_endpgm");

  // ---- kernel descriptor + metadata ------------------------------------------
  const uint32_t next_vgpr = std::max<uint32_t>(V_BASE + 2 * std::max<uint32_t>(prog.n_reg_used, 1), 8);
  const uint32_t accum = (next_vgpr + 3) & ~3u;
  os << "\t.section\t.rodata,\"a\",@progbits\n\t.p2align\t6, 0x0\n\t.amdhsa_kernel " << kname << "\n";
  os << "\t\t.amdhsa_group_segment_fixed_size " << lds_bytes << "\n";
  os << "\t\t.amdhsa_private_segment_fixed_size 0\n\t\t.amdhsa_kernarg_size 72\n\t\t.amdhsa_user_sgpr_count 2\n";
  os << "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1\n\t\t.amdhsa_system_sgpr_workgroup_id_x 1\n";
  os << "\t\t.amdhsa_system_vgpr_workitem_id 0\n";
  const uint32_t n_agpr = 2 * prog.n_acc_used;
  os << "\t\t.amdhsa_next_free_vgpr " << (accum + n_agpr) << "\n\t\t.amdhsa_next_free_sgpr " << S_END << "\n";
  os << "\t\t.amdhsa_accum_offset " << accum << "\n\t\t.amdhsa_reserve_vcc 1\n";
  os << "\t\t.amdhsa_float_round_mode_32 0\n\t\t.amdhsa_float_round_mode_16_64 0\n";
  os << "\t\t.amdhsa_float_denorm_mode_32 3\n\t\t.amdhsa_float_denorm_mode_16_64 3\n";
  os << "\t\t.amdhsa_dx10_clamp 1\n\t\t.amdhsa_ieee_mode 1\n";
  os << "\t.end_amdhsa_kernel\n\t.text\n";
  os << "\t.amdgpu_metadata\n---\namdhsa.kernels:\n  - .agpr_count: " << n_agpr << "\n    .args:\n";
  const char *kinds[9] = {"global_buffer", "by_value", "by_value", "global_buffer", "by_value", "by_value",
                          "global_buffer", "by_value", "by_value"};
  for (int i = 0; i < 9; ++i) {
    os << "      - .offset: " << i * 8 << "\n        .size: 8\n        .value_kind: " << kinds[i] << "\n";
    if (std::strcmp(kinds[i], "global_buffer") == 0) os << "        .address_space: global\n";
  }
  os << "    .group_segment_fixed_size: " << lds_bytes << "\n    .kernarg_segment_align: 8\n    .kernarg_segment_size: 72\n";
  os << "    .max_flat_workgroup_size: 64\n    .name: " << kname << "\n    .private_segment_fixed_size: 0\n";
  os << "    .sgpr_count: " << (S_END + 6) << "\n    .sgpr_spill_count: 0\n    .symbol: " << kname << ".kd\n";
  os << "    .uniform_work_group_size: 1\n    .uses_dynamic_stack: false\n    .vgpr_count: " << (accum + n_agpr)
     << "\n    .vgpr_spill_count: 0\n    .wavefront_size: 64\n";
  os << "amdhsa.target: amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\t.end_amdgpu_metadata\n";
  (void)p;
  return os.str();
}

}  // namespace fdg
