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
// Kernel arguments (all 8 bytes): leaf, ss, ls, root, rs, rk, ws, B, nwg, weight [, leaf2, ls2, -kF^2, beta, -beta, lambda:
//   Monte-Carlo kernels, whose input columns n_k.. (the times) are leaf2[b*ss + (i - n_k)*ls2] and whose formulas read the
//   four physical parameters from SGPRs], lts, rts
//   lts / rts: element distance between the first samples of consecutive tiles of the leaf / root arrays.  A plain strided
//   matrix has lts = T ss (T = samples per tile, 64 or 128); a TILE-MAJOR batch leaf[b / 64][i][b % 64] (a Julia
//   Array{Float64,3}(64, L, cld(B, 64))) has ss = 1, ls = 64, lts = 64 L: the 64 samples x L leaves of a tile are one
//   contiguous block that its wave streams front to back.
//   (the _acc variant keeps acc_k += w * root_k in registers and writes one partial per wave and root to `root`)
//   leaf value i of sample b: leaf[b*ss + i*ls]; root k: root[b*rs + k*rk]
#include <algorithm>
#include <cctype>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>

#include "fdg_opt.h"

namespace fdg {
namespace {

constexpr int V_LANE8 = 1, V_LEAFOFF = 2, V_ROOTOFF = 3, V_TMP = 4, V_BASE = 6;
constexpr int S_LEAF = 4, S_SS = 6, S_LS = 8, S_ROOT = 10, S_RS = 12, S_RK = 14, S_WS = 16, S_B = 18, S_NWG = 20;
constexpr int S_TILE = 22, S_NTILES = 23, S_LT = 24, S_RT = 26, S_PANEL = 28, S_LS8 = 30, S_RK8 = 32, S_A = 34,
              S_T = 36, S_C = 38, S_X = 40;  // S_X.. : scratch (4)
constexpr int S_WGT = 44;
constexpr int S_RTS = 44;    // root tile stride (kernels that write roots; the accumulating ones keep the weight pointer here)
constexpr int S_LTS = 46;    // leaf tile stride
constexpr int S_LP = 48;     // running pointer: column of the most recently loaded leaf in this tile
constexpr int S_DELTA = 50;  // up to N_DELTA pairs: (leaf stride in bytes) x the most frequent index steps between consecutive loads
constexpr int N_DELTA = 6;
constexpr int S_POOL = S_DELTA + 2 * N_DELTA;   // constants of the graph (edge factors without an inline encoding), loaded once per wave
constexpr int N_POOL = 20;   // (16 are used by programs without leaf formulas, 13 by Monte-Carlo kernels)
constexpr int S_END = S_POOL + 2 * N_POOL;
// Monte-Carlo kernels (two more arguments: the times' base and column stride) keep these in the pool's last three pairs
constexpr int N_POOL_MC = 13;
constexpr int S_PARAM = S_POOL + 2 * N_POOL_MC;   // the four physical parameters of MOp::param (kernel arguments 13-16)
constexpr int S_LEAF2 = S_PARAM + 2 * MC_N_PARAM, S_LS82 = S_LEAF2 + 2, S_LT2 = S_LEAF2 + 4;
static_assert(S_LT2 + 2 == S_END && S_PARAM % 4 == 0 && S_LEAF2 % 4 == 0, "SGPR map");

// exp(x), x <= 0 in practice: n = rint(x log2 e), r = x - n ln2 (two-part), exp(r) by a degree-11 polynomial
// (Chebyshev-node interpolant of exp on |r| <= 0.3467, coefficients rounded to double: 1.6e-17 relative, computed with
// 60-digit arithmetic), scaled by 2^n with v_ldexp_f64 (flushes to 0 / denormals correctly far below)
const double kLog2e = 0x1.71547652b82fep+0, kLn2Hi = 0x1.62e42fefa39efp-1, kLn2Lo = 0x1.abc9e3b39803fp-56;
constexpr int kExpDeg = 11;
const double kExpC[kExpDeg + 1] = {0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000011p-1, 0x1.555555555555ap-3,
                                   0x1.555555554f0a5p-5, 0x1.111111110f218p-7, 0x1.6c16c18804745p-10, 0x1.a01a01b148c00p-13,
                                   0x1.a019919593233p-16, 0x1.71ddf5667e394p-19, 0x1.28b41ab9f014bp-22, 0x1.af63371ef88d9p-26};


// ---- gfx950 required wait states between dependent instructions ---------------------------------------------------
// Hand-written assembly gets no help from a compiler's hazard recogniser, so the pairs this emitter can produce are
// kept in ONE table: Emit::ins() consults it for every instruction and puts the missing `s_nop` in front, and
// fdg_isa_check_hazards() re-parses a finished listing against the same table (tests/test_isa_hazards.py runs it over
// every prebuilt kernel).  A wait state is one instruction issued in between; `s_nop N` counts N + 1.
// Sources: the gfx90a/gfx940 hazard rules of LLVM's AMDGPU back end, and where noted our own measurements on MI355X
// (a violated pair showed up as a stale register read).
enum HzRes : uint8_t { HZ_VGPR, HZ_SGPR /* incl. vcc */ };
enum HzProd : uint8_t {
  HP_TRANS,        // VALU transcendental (v_rcp/v_rsq/v_sqrt/v_exp/v_log/v_sin/v_cos) writes a VGPR
  HP_VALU_SGPR,    // VALU writes an SGPR or VCC (v_cmp*, v_div_scale, v_readfirstlane, carry-out)
  HP_VALU_VGPR,    // any VALU writes a VGPR
  HP_STORE_WIDE,   // VMEM / DS store whose data operand is wider than 64 bits reads a VGPR
  HP_SALU_SGPR,    // SALU writes an SGPR
};
enum HzCons : uint8_t {
  HC_VALU_READ,    // VALU reads the register as an operand (VGPR) / as mask, carry-in or constant (SGPR, VCC)
  HC_DIV_FMAS,     // v_div_fmas_* reads VCC
  HC_VMEM_SADDR,   // VMEM / SMEM instruction reads the SGPR (address, offset)
  HC_LANE_READ,    // v_readlane / v_readfirstlane reads the VGPR
  HC_VALU_WRITE,   // VALU overwrites the VGPR
  HC_LDS_DIRECT,   // global_load_lds_* reads M0 (its LDS base)
};
struct HzRule { HzProd prod; HzCons cons; HzRes res; int wait; const char *what; };
const HzRule kHazards[] = {
    {HP_TRANS, HC_VALU_READ, HZ_VGPR, 2, "trans result -> VALU read (gfx940: not forwarded; measured: stale read with 0)"},
    {HP_VALU_SGPR, HC_VALU_READ, HZ_SGPR, 2, "VALU write of SGPR/VCC -> VALU read as mask/carry/constant (gfx940; measured with v_cmp -> v_cndmask)"},
    {HP_VALU_SGPR, HC_DIV_FMAS, HZ_SGPR, 4, "VALU write of VCC -> v_div_fmas"},
    {HP_VALU_SGPR, HC_VMEM_SADDR, HZ_SGPR, 5, "VALU write of SGPR -> VMEM/SMEM read of it"},
    {HP_VALU_VGPR, HC_LANE_READ, HZ_VGPR, 2, "VALU write of VGPR -> v_readlane/v_readfirstlane (measured: stale read with 0)"},
    {HP_STORE_WIDE, HC_VALU_WRITE, HZ_VGPR, 2, "store data wider than 64 bits -> VALU overwrite of the data registers"},
    {HP_SALU_SGPR, HC_VMEM_SADDR, HZ_SGPR, 0, "SALU write of SGPR -> VMEM read of it: interlocked by hardware (every leaf load does this back to back)"},
    {HP_SALU_SGPR, HC_LDS_DIRECT, HZ_SGPR, 1, "SALU write of M0 -> LDS-direct load (global_load_lds_*)"},
};
constexpr int HZ_MAX_WAIT = 5;

// registers are numbered: VGPR v -> v, AGPR a -> 512 + a, SGPR s -> 1024 + s, vcc -> 1024 + 106/107, exec -> 1024 + 126/127
constexpr int HR_AGPR = 512, HR_SGPR = 1024, HR_VCC = 1024 + 106, HR_EXEC = 1024 + 126, HR_M0 = 1024 + 124;
struct HzInst {
  std::string op;
  bool valu = false, salu = false, vmem = false, ds = false, smem = false, trans = false, lane_read = false, div_fmas = false, store = false, lds_direct = false;
  int nop = 0;                  // s_nop: wait states it provides
  std::vector<int> wr, rd;      // registers written / read (numbering above)
  std::vector<int> store_wide;  // data registers of a wide store
};
static bool hz_starts(const std::string &s, const char *p) { return s.compare(0, std::strlen(p), p) == 0; }
// "v[4:5]" "-v7" "s[10:11]" "a3" "vcc" "exec" "0x12" "1.0" "off" ...  -> register range (first, count), count 0 = not a register
static void hz_parse_reg(std::string t, int &first, int &n) {
  first = 0; n = 0;
  while (!t.empty() && (t[0] == '-' || t[0] == '|' || t[0] == ' ')) t.erase(0, 1);
  while (!t.empty() && (t.back() == '|' || t.back() == ' ')) t.pop_back();
  if (t == "vcc") { first = HR_VCC; n = 2; return; }
  if (t == "vcc_lo") { first = HR_VCC; n = 1; return; }
  if (t == "vcc_hi") { first = HR_VCC + 1; n = 1; return; }
  if (t == "exec") { first = HR_EXEC; n = 2; return; }
  if (t == "m0") { first = HR_M0; n = 1; return; }
  if (t.size() < 2 || (t[0] != 'v' && t[0] != 's' && t[0] != 'a')) return;
  const int base = t[0] == 'v' ? 0 : (t[0] == 'a' ? HR_AGPR : HR_SGPR);
  if (t[1] == '[') {
    int a = 0, b = 0;
    if (std::sscanf(t.c_str() + 2, "%d:%d", &a, &b) == 2 && b >= a) { first = base + a; n = b - a + 1; }
    return;
  }
  if (t[1] < '0' || t[1] > '9') return;
  first = base + std::atoi(t.c_str() + 1); n = 1;
}
static HzInst hz_decode(const std::string &line) {
  HzInst I;
  size_t p = 0;
  while (p < line.size() && (line[p] == ' ' || line[p] == '\t')) ++p;
  size_t q = p;
  while (q < line.size() && line[q] != ' ' && line[q] != '\t') ++q;
  I.op = line.substr(p, q - p);
  std::vector<std::string> opnd;
  {
    std::string rest = line.substr(q), cur;
    int depth = 0;
    for (char c : rest) {
      if (c == '[') depth++;
      if (c == ']') depth--;
      if (c == ',' && depth == 0) { opnd.push_back(cur); cur.clear(); } else cur.push_back(c);
    }
    if (!cur.empty()) opnd.push_back(cur);
    for (std::string &o : opnd) {           // strip blanks and trailing modifiers ("offset:8", "glc")
      while (!o.empty() && (o[0] == ' ' || o[0] == '\t')) o.erase(0, 1);
      const size_t sp = o.find_first_of(" \t");
      if (sp != std::string::npos) o.erase(sp);
    }
  }
  auto add = [](std::vector<int> &v, const std::string &t) { int f, n; hz_parse_reg(t, f, n); for (int i = 0; i < n; ++i) v.push_back(f + i); };
  const std::string &op = I.op;
  if (op == "s_nop") { I.salu = true; I.nop = (opnd.empty() ? 0 : std::atoi(opnd[0].c_str())) + 1; return I; }
  if (hz_starts(op, "s_waitcnt") || op == "s_endpgm" || op == "s_barrier" || hz_starts(op, "s_cbranch") || op == "s_branch") { I.salu = true; return I; }
  if (hz_starts(op, "s_load") || hz_starts(op, "s_buffer_load")) {
    I.smem = true;
    if (!opnd.empty()) add(I.wr, opnd[0]);
    for (size_t i = 1; i < opnd.size(); ++i) add(I.rd, opnd[i]);
    return I;
  }
  if (hz_starts(op, "s_")) {
    I.salu = true;
    const bool nodst = hz_starts(op, "s_cmp") || hz_starts(op, "s_setpc") || hz_starts(op, "s_bitcmp");
    for (size_t i = 0; i < opnd.size(); ++i) add((i == 0 && !nodst) ? I.wr : I.rd, opnd[i]);
    if (hz_starts(op, "s_addc") || hz_starts(op, "s_subb") || hz_starts(op, "s_cselect")) {}   // SCC is not tracked (SALU only)
    return I;
  }
  if (hz_starts(op, "global_load") || hz_starts(op, "global_store") || hz_starts(op, "buffer_") || hz_starts(op, "flat_")) {
    I.vmem = true;
    I.store = op.find("store") != std::string::npos;
    if (op.find("_lds_") != std::string::npos) {      // LDS-direct: no VGPR destination, M0 is the LDS base
      I.lds_direct = true;
      I.rd.push_back(HR_M0);
      for (size_t i = 0; i < opnd.size(); ++i) add(I.rd, opnd[i]);
      return I;
    }
    const bool wide = op.find("dwordx3") != std::string::npos || op.find("dwordx4") != std::string::npos;
    for (size_t i = 0; i < opnd.size(); ++i) {
      if (!I.store && i == 0) add(I.wr, opnd[i]);
      else { add(I.rd, opnd[i]); if (I.store && i == 1 && wide) add(I.store_wide, opnd[i]); }
    }
    return I;
  }
  if (hz_starts(op, "ds_")) {
    I.ds = true;
    I.store = hz_starts(op, "ds_write");
    const bool wide = op.find("b96") != std::string::npos || op.find("b128") != std::string::npos;
    for (size_t i = 0; i < opnd.size(); ++i) {
      if (!I.store && i == 0) add(I.wr, opnd[i]);
      else { add(I.rd, opnd[i]); if (I.store && i == 1 && wide) add(I.store_wide, opnd[i]); }
    }
    return I;
  }
  if (hz_starts(op, "v_")) {
    I.valu = true;
    static const char *trans[] = {"v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_"};
    for (const char *t : trans) if (hz_starts(op, t)) I.trans = true;
    I.lane_read = hz_starts(op, "v_readlane") || hz_starts(op, "v_readfirstlane");
    I.div_fmas = hz_starts(op, "v_div_fmas");
    size_t ndst = 1;
    if (hz_starts(op, "v_div_scale") || ((hz_starts(op, "v_add_co") || hz_starts(op, "v_sub_co") || hz_starts(op, "v_addc_co") || hz_starts(op, "v_subb_co")) && opnd.size() >= 4)) ndst = 2;
    if (hz_starts(op, "v_cmpx")) { ndst = 0; I.wr.push_back(HR_EXEC); I.wr.push_back(HR_EXEC + 1); }
    for (size_t i = 0; i < opnd.size(); ++i) add(i < ndst ? I.wr : I.rd, opnd[i]);
    if (I.div_fmas) { I.rd.push_back(HR_VCC); I.rd.push_back(HR_VCC + 1); }
    return I;
  }
  return I;   // directives, labels: not instructions
}
static bool hz_is_inst(const std::string &line) {
  size_t p = 0;
  while (p < line.size() && (line[p] == ' ' || line[p] == '\t')) ++p;
  if (p >= line.size() || line[p] == '.' || line[p] == ';' || line[p] == '/' || line[p] == '-') return false;
  const size_t q = line.find_first_of(" \t", p);
  const std::string w = line.substr(p, q == std::string::npos ? std::string::npos : q - p);
  if (!w.empty() && w.back() == ':') return false;   // label
  return hz_starts(w, "s_") || hz_starts(w, "v_") || hz_starts(w, "ds_") || hz_starts(w, "global_") || hz_starts(w, "buffer_") || hz_starts(w, "flat_");
}

// Sliding window over the last HZ_MAX_WAIT wait states.
struct HzTracker {
  struct Past { HzInst inst; int age; };     // age = wait states issued since (0 = the next instruction follows immediately)
  std::vector<Past> past;
  // wait states missing before `I` may issue; `why` names the rule
  int missing(const HzInst &I, const char **why = nullptr) const {
    int need = 0;
    auto has = [](const std::vector<int> &v, int r) { return std::find(v.begin(), v.end(), r) != v.end(); };
    for (const Past &P : past) {
      for (const HzRule &R : kHazards) {
        if (R.wait <= P.age) continue;
        const std::vector<int> *pr = nullptr;
        switch (R.prod) {
          case HP_TRANS: if (P.inst.trans) pr = &P.inst.wr; break;
          case HP_VALU_SGPR: if (P.inst.valu) pr = &P.inst.wr; break;
          case HP_VALU_VGPR: if (P.inst.valu) pr = &P.inst.wr; break;
          case HP_STORE_WIDE: if (!P.inst.store_wide.empty()) pr = &P.inst.store_wide; break;
          case HP_SALU_SGPR: if (P.inst.salu) pr = &P.inst.wr; break;
        }
        if (!pr) continue;
        const std::vector<int> *cr = nullptr;
        switch (R.cons) {
          case HC_VALU_READ: if (I.valu && !I.div_fmas) cr = &I.rd; break;
          case HC_DIV_FMAS: if (I.div_fmas) cr = &I.rd; break;
          case HC_VMEM_SADDR: if (I.vmem || I.smem) cr = &I.rd; break;
          case HC_LANE_READ: if (I.lane_read) cr = &I.rd; break;
          case HC_VALU_WRITE: if (I.valu) cr = &I.wr; break;
          case HC_LDS_DIRECT: if (I.lds_direct) cr = &I.rd; break;
        }
        if (!cr) continue;
        for (int r : *pr) {
          const bool sg = r >= HR_SGPR;
          if ((R.res == HZ_SGPR) != sg) continue;
          if (R.cons == HC_LDS_DIRECT && r != HR_M0) continue;
          if (has(*cr, r)) { if (R.wait - P.age > need) { need = R.wait - P.age; if (why) *why = R.what; } break; }
        }
      }
    }
    return need;
  }
  void issue(const HzInst &I) {
    const int ws = I.nop ? I.nop : 1;
    for (Past &P : past) P.age += ws;
    past.erase(std::remove_if(past.begin(), past.end(), [](const Past &P) { return P.age >= HZ_MAX_WAIT; }), past.end());
    if (!I.nop && (I.valu || !I.store_wide.empty())) past.push_back(Past{I, 0});
    else if (!I.nop && I.salu && !I.wr.empty()) past.push_back(Past{I, 0});
  }
  void reset() { past.clear(); }
};

// ---- encoded size of a printed instruction -----------------------------------------------------------------------------------
// Round 5 (tools/ubench/valu_align.hip, profiles/r05_ubench_valu_align.txt): an 8-byte instruction that straddles a 64-byte line of the
// instruction stream costs a lone wave about 8 cycles (two issue slots) -- a stream displaced by 4 bytes runs its fp64 operations at 5.08
// instead of 4.13 cycles; an s_nop costs 4.  The printer therefore keeps count of where it is within the line (every kernel starts at
// .p2align 8 and every instruction goes through Emit::ins) and puts one s_nop in front of an instruction that would straddle.  Only the
// forms this file prints are classified: VOP1/VOP2 carry their _e32 / _e64 suffix, everything else starting with v_ is VOP3;
// tests/test_isa_alignment.py holds the classification against the assembler's own listing.
static bool isa_operand_is_literal(std::string t) {
  while (!t.empty() && (t.front() == ' ' || t.front() == '\t')) t.erase(t.begin());
  while (!t.empty() && (t.back() == ' ' || t.back() == '\t')) t.pop_back();
  if (t.empty()) return false;
  std::string u = t;
  if (u[0] == '-' || u[0] == '|') u.erase(u.begin());
  if (!u.empty() && (u[0] == 's' || u[0] == 'v' || u[0] == 'a') && u.size() > 1 && (std::isdigit((unsigned char)u[1]) || u[1] == '[')) return false;
  for (const char *r : {"vcc", "exec", "m0", "scc", "off", "null", "vmcnt", "lgkmcnt", "expcnt", "src_"}) if (u.compare(0, std::strlen(r), r) == 0) return false;
  char *end = nullptr;
  const long long v = std::strtoll(t.c_str(), &end, 0);
  if (end && *end == 0 && end != t.c_str()) return v < -16 || v > 64;
  const double f = std::strtod(t.c_str(), &end);
  if (end && *end == 0 && end != t.c_str()) { const double a = std::fabs(f); return !(a == 0.5 || a == 1.0 || a == 2.0 || a == 4.0); }
  return true;                                   // an expression over symbols
}
static int isa_size(const std::string &s) {
  const size_t e = s.find_first_of(" \t");
  const std::string mn = s.substr(0, e);
  auto starts = [&](const char *p) { return mn.compare(0, std::strlen(p), p) == 0; };
  auto ends = [&](const char *p) { const size_t n = std::strlen(p); return mn.size() >= n && mn.compare(mn.size() - n, n, p) == 0; };
  if (starts("ds_") || starts("global_") || starts("buffer_") || starts("flat_") || starts("scratch_")) return 8;
  if (starts("s_load_") || starts("s_buffer_") || starts("s_store_") || starts("s_dcache") || mn == "s_memtime" || mn == "s_memrealtime") return 8;
  bool lit = false;
  if (e != std::string::npos) {
    size_t a = e;
    while (a <= s.size()) {
      size_t c = s.find(',', a);
      if (c == std::string::npos) c = s.size();
      if (isa_operand_is_literal(s.substr(a, c - a))) lit = true;
      a = c + 1;
    }
  }
  if (starts("s_")) {
    for (const char *q : {"s_nop", "s_waitcnt", "s_barrier", "s_endpgm", "s_branch", "s_cbranch", "s_sleep", "s_setprio", "s_sendmsg", "s_sethalt", "s_trap",
                          "s_icache_inv", "s_movk", "s_addk", "s_mulk", "s_cmpk", "s_cmovk", "s_getreg", "s_call_b64"})
      if (starts(q)) return 4;
    return lit ? 8 : 4;
  }
  if (starts("v_")) {
    if (ends("_e64")) return 8;
    if (ends("_e32") || mn == "v_readfirstlane_b32" || mn == "v_nop") return lit ? 8 : 4;
    return 8;
  }
  return 4;
}

struct Emit {
  std::ostringstream os;
  HzTracker hz;
  uint32_t off = 0;           // bytes printed since the kernel's .p2align 8
  int align = 1;              // option FDG_ISA_ALIGN: 0 print as rounds 1-4 did, 1 no 8-byte instruction straddles, 2 no fp64 / VOP3 one does
  bool pad_ok = true;         // false inside the cooperative / pooled kernels: their waves meet at barriers, and the pads cost gv_ver4_4's pooled kernel 12 %
                              // (profiles/r05_log_align_shift.txt)
  int shift = 0;              // dev: FDG_ISA_SHIFT=n puts n s_nop at the head of every kernel (how much does the mere position matter?)
  uint64_t n_align_nop = 0;
  bool streaming = false;     // the kernel being printed is the variant for line-aligned batches: non-temporal leaf loads and root stores
  long nt_dist = -1;          // streaming kernels of long programs: a re-loaded leaf's earlier load is non-temporal too when the re-load is this many leaf loads away (-1: never)
  uint64_t n_auto_nop = 0;
  uint64_t vm_issued = 0, lg_issued = 0, vm_done = 0, lg_done = 0;
  bool no_vm_wait = false;    // FDG_ISA_DEBUG=novmwait (timing experiment, results are garbage): no wait for a load's data -- what a wave whose loads always
                              // arrived in time would run at (the loads are still issued)
  uint64_t vm_slack = 0;      // FDG_ISA_DEBUG=noackwait (timing experiment, results may be garbage): waits let this many more operations stay
                              // outstanding -- the tile's root stores -- to see what a wave that never waits for store acknowledgements would run at
  // pending[reg] = (kind 0 none / 1 vm / 2 lgkm, seq)
  std::vector<std::pair<uint8_t, uint64_t>> pend;
  std::vector<uint64_t> pend_acc;      // [AGPR pair] vm sequence number of the leaf load that lands in it (0: none outstanding)

  std::string klabel;         // label of the kernel being printed (`off` counts from it)
  bool check_off = false;     // dev: FDG_ISA_CHECK_OFF=1 makes the assembler itself hold `off` against the location counter every 32 instructions
  uint64_t n_ins = 0;
  // every instruction goes through here: the wait states the hazard table demands are put in front of it
  void ins(const std::string &s) {
    if (check_off && !klabel.empty() && (n_ins++ & 31u) == 0)
      os << ".if (. - " << klabel << ") != " << off << "\n.error \"Emit::off is wrong in front of: " << s << "\"\n.endif\n";
    const HzInst I = hz_decode(s);
    const int need = hz.missing(I);
    if (need > 0) { os << "\ts_nop " << (need - 1) << "\n"; HzInst N; N.salu = true; N.nop = need; hz.issue(N); n_auto_nop++; off += 4; }
    const int sz = isa_size(s);
    if (align && pad_ok && sz == 8 && (off & 63u) == 60u && (align == 1 || s.compare(0, 2, "v_") == 0)) { os << "\ts_nop 0\n"; HzInst N; N.salu = true; N.nop = 1; hz.issue(N); n_align_nop++; off += 4; }
    off += (uint32_t)sz;
    hz.issue(I);
    os << "\t" << s << "\n";
  }
  // a label: control flow may arrive from elsewhere, but every path into our labels ends in SALU branches issued long
  // after the last VALU/VMEM instruction of the previous block, except the tile loop's back edge (handled by the nops
  // emitted in front of s_setpc)
  void label(const std::string &l) { os << l << ":\n"; }
  std::string vr(uint32_t r) const {
    char b[32];
    std::snprintf(b, sizeof b, "v[%u:%u]", V_BASE + 2 * r, V_BASE + 2 * r + 1);
    return b;
  }
  void wait_reg(uint32_t r) {
    auto &p = pend[r];
    if (p.first == 1 && no_vm_wait) { p.first = 0; return; }
    if (p.first == 1 && p.second > vm_done) {
      uint64_t n = vm_issued - p.second + vm_slack;            // ops issued after it may stay outstanding
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
  void wait_vm(uint64_t seq) {
    if (seq <= vm_done || no_vm_wait) return;
    uint64_t n = std::min<uint64_t>(vm_issued - seq + vm_slack, 63);
    ins("s_waitcnt vmcnt(" + std::to_string(n) + ")");
    vm_done = std::max(std::max(vm_done, vm_issued - n), seq);
  }
  void wait_lg_seq(uint64_t seq) {       // LDS operations complete in order: everything up to `seq` has returned
    if (seq <= lg_done) return;
    const uint64_t n = std::min<uint64_t>(lg_issued - seq, 15);
    ins("s_waitcnt lgkmcnt(" + std::to_string(n) + ")");
    lg_done = std::max(std::max(lg_done, lg_issued - n), seq);
    for (auto &p : pend) if (p.first == 2 && p.second <= lg_done) p.first = 0;
  }
  void drain() {
    ins("s_waitcnt vmcnt(0) lgkmcnt(0)");
    vm_done = vm_issued;
    lg_done = lg_issued;
    for (auto &p : pend) p.first = 0;
    for (auto &p : pend_acc) p = 0;
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

struct KernelMeta { std::string name; uint32_t lds_bytes, accum, n_agpr; int n_args, n_sgpr; uint32_t wg = 64; };

// Prints one kernel.  W = samples per lane: 1 (64-sample tiles, 8-byte accesses) or 2 (128-sample
// tiles: a value is two doubles in four VGPRs, every memory access is 16 bytes per lane -- the
// wide-access form HBM-bound graphs want; needs sample stride 1 and full tiles, the host runs the
// remainder through the W = 1 kernel).
//
// rm_bufs > 0: the row-major variant (compile_Python's [B, L] input, leaf stride 1).  A wave's 64 samples are 64 rows of
// the matrix; chunks of RM_CHUNK consecutive leaves of those rows (64 x 128 bytes) are brought into `rm_bufs` LDS staging
// buffers by LDS-direct loads (global_load_lds_dwordx4: eight instructions per chunk, every 128-byte row segment one
// coalesced line, no VGPR in between), and a leaf's first use reads lane = row from there (ds_read_b64).  The LDS image
// of a chunk is lane-linear by construction of the instruction, so the bank swizzle sits in the SOURCE address: lane l
// of instruction n fetches piece (l % 8) ^ (l / 8) of row 8 n + l / 8, which puts piece j of row r at
// r * 128 + ((j ^ (r % 8)) * 16): a column read by 64 rows touches every bank pair twice instead of one pair 64 times.
// Which chunk is resident when is planned here from the allocated program (Belady over the buffers); a leaf used again
// long after its chunk left is gathered straight from memory (eight bytes per lane, rare).  Only full 64-row tiles.
constexpr uint32_t RM_CHUNK = 16, RM_BUF_BYTES = 64 * RM_CHUNK * 8, RM_WINDOW = 800;
// chunk c = leaves 16 c .. 16 c + 15; a last partial chunk is shifted back to end at the last leaf (never reads past a row)
struct RmFetch { uint32_t chunk, buf; };
static void rm_plan(const Lowered &p, const OptProgram &prog, uint32_t rm_bufs, std::vector<std::vector<RmFetch>> &rm_fetch,
                    std::vector<int> &rm_ld_buf) {
  const uint32_t rm_full = p.L / RM_CHUNK, rm_tail = (p.L % RM_CHUNK) ? 1u : 0u, n_chunk = rm_full + rm_tail;
  auto rm_chunk_of = [&](uint32_t leaf) { return leaf < rm_full * RM_CHUNK ? leaf / RM_CHUNK : rm_full; };
  // Round 6 (OptParams::rm_pair, four buffers): chunks are fetched in PAIRS -- chunks 2u and 2u + 1 back to back into two adjacent buffers.  A
  // 128-byte row segment of a row that is not a whole number of lines straddles two cache lines, and the line it shares with the next chunk is
  // requested again when that chunk is fetched -- a chunk's worth of computing later, by which time the XCD's L2 has turned over for a quarter of
  // them: the chunked variant moves 1.25 x the matrix, and its fetch stream ALONE runs at the memory system's ceiling for that traffic
  // (profiles/r06_log_sweep_m.txt).  Fetched together, the two halves of a pair ask for their shared line within nanoseconds of each other.
  const uint32_t G = (prog.params.rm_pair && rm_bufs >= 4 && rm_bufs % 2 == 0) ? 2u : 1u;
  const uint32_t n_unit = (n_chunk + G - 1) / G, n_ubuf = rm_bufs / G;
  const size_t n_ops = prog.ops.size();
  rm_fetch.assign(n_ops + 1, {});
  rm_ld_buf.assign(n_ops, -1);
  std::vector<std::vector<size_t>> uses(n_unit);
  for (size_t q = 0; q < n_ops; ++q) if (prog.ops[q].kind == M_LD_LEAF) uses[rm_chunk_of(prog.ops[q].a) / G].push_back(q);
  std::vector<size_t> cursor(uses.size(), 0);
  std::vector<int64_t> resident(n_ubuf, -1);
  std::vector<size_t> free_from(n_ubuf, 0);      // op index from which the buffer (pair) may be filled again
  for (size_t q = 0; q < n_ops; ++q) {
    if (prog.ops[q].kind != M_LD_LEAF) continue;
    const uint32_t c = rm_chunk_of(prog.ops[q].a), u = c / G;
    cursor[u]++;                                   // uses[u][cursor[u]..] are the later ones
    int b = -1;
    for (uint32_t k = 0; k < n_ubuf; ++k) if (resident[k] == (int64_t)u) b = (int)k;
    if (b < 0) {
      // a chunk is worth a buffer (and eight loads) when at least three of its leaves are read within the next
      // RM_WINDOW ops; stragglers -- a value used again long after its neighbours -- are gathered from memory
      size_t soon = 1;
      for (size_t k = cursor[u]; k < uses[u].size() && uses[u][k] <= q + RM_WINDOW; ++k) soon++;
      if (soon < 3) continue;
      // victim: an empty buffer, else the resident chunk whose next use is farthest (none at all first)
      size_t far = 0;
      for (uint32_t k = 0; k < n_ubuf; ++k) {
        size_t nu;
        if (resident[k] < 0) nu = std::numeric_limits<size_t>::max();
        else { const auto &uu = uses[(size_t)resident[k]]; const size_t cu = cursor[(size_t)resident[k]]; nu = cu < uu.size() ? uu[cu] : std::numeric_limits<size_t>::max() - 1; }
        if (b < 0 || nu > far) { b = (int)k; far = nu; }
      }
      for (uint32_t k = 0; k < G && G * u + k < n_chunk; ++k)
        rm_fetch[std::min(free_from[(size_t)b], q)].push_back(RmFetch{G * u + k, G * (uint32_t)b + k});
      resident[(size_t)b] = u;
    }
    rm_ld_buf[q] = (int)(G * (uint32_t)b + c % G);
    free_from[(size_t)b] = q + 1;
  }
}

// One wave's section of the cooperative kernel (emit_coop below): no kernel header or descriptor of its own; lane = thread id
// & 63; private LDS slots behind the shared ones (addressed through their own base register); the wave's panel inside the
// workgroup's; M_SEND / M_RECV / M_BARRIER.
struct CoopSec { uint32_t wave, n_shared, priv_base_bytes, panel_wg_bytes, panel_prefix_bytes; bool pooled = false; uint32_t pool_unit = 1;
                 uint32_t slack = 0, flag_base_bytes = 0, n_wave = 4; };     // slack > 0: progress words instead of s_barrier between the epochs of a tile (fdg_opt.h: CoopProgram::slack)
// rl: the row-major variant for CONTIGUOUS rows (sample stride == L: compile_Python's [B, L] exactly) of graphs whose tile fits the LDS: a
// tile's 64 rows are one block of 512 L bytes, streamed linearly into an LDS image by LDS-direct loads (1 KB per instruction, every cache
// line of the matrix requested exactly once, non-temporal), and leaf i of lane = row r is read from image[r * 8 L + 8 i].
static KernelMeta emit_kernel(Emit &E, const Lowered &p, const OptProgram &prog, const std::string &kname, int W, bool accumulate = false,
                              uint32_t rm_bufs = 0, const CoopSec *cs = nullptr, bool rl = false) {
  E.vm_issued = E.lg_issued = E.vm_done = E.lg_done = 0;
  E.pend.assign(std::max<uint32_t>(prog.n_reg_used, 1), {0, 0});
  const uint32_t SLOT = 512u * W;                 // bytes of one LDS / panel slot of a wave
  const uint32_t stage_base = (prog.n_lds_used * SLOT + 1023u) & ~1023u;
  const uint32_t rl_image = rl ? ((512u * p.L + 1023u) & ~1023u) : 0u;
  const uint32_t lds_bytes = rl ? stage_base + rl_image : (rm_bufs ? stage_base + rm_bufs * RM_BUF_BYTES : prog.n_lds_used * SLOT);
  const uint32_t panel_bytes_per_wave = std::max<uint32_t>(prog.n_mem_used, 1) * SLOT;
  const int RW = 2 * W;                           // VGPRs per value
  const int TSH = W == 2 ? 7 : 6;                 // log2(samples per tile)
  const std::string sfx = "_" + kname;
  std::ostringstream &os = E.os;
  if (!cs) {
    os << "\t.text\n\t.protected\t" << kname << "\n\t.globl\t" << kname << "\n\t.p2align\t8\n\t.type\t" << kname << ",@function\n";
    os << kname << ":\n";
    E.off = 0;
    E.klabel = kname;
    E.pad_ok = true;
    for (int i = 0; i < E.shift; ++i) E.ins("s_nop 0");
  } else {
    os << ".Lsec" << sfx << ":\n";
  }
  auto S = [](int r) { return "s" + std::to_string(r); };
  auto S2 = [](int r) { return "s[" + std::to_string(r) + ":" + std::to_string(r + 1) + "]"; };
  auto V = [](int r) { return "v" + std::to_string(r); };
  auto vlo = [&](uint32_t r) { const int b = V_BASE + RW * r; return "v[" + std::to_string(b) + ":" + std::to_string(b + 1) + "]"; };
  auto vhi = [&](uint32_t r) { const int b = V_BASE + RW * r + 2; return "v[" + std::to_string(b) + ":" + std::to_string(b + 1) + "]"; };
  auto vall = [&](uint32_t r) { const int b = V_BASE + RW * r; return "v[" + std::to_string(b) + ":" + std::to_string(b + RW - 1) + "]"; };
  const std::string LD = W == 2 ? "global_load_dwordx4 " : "global_load_dwordx2 ";
  const std::string ST = W == 2 ? "global_store_dwordx4 " : "global_store_dwordx2 ";
  // cache policy of the leaf stream (experiment knob): FDG_ISA_LEAF_POLICY="nt" / "sc1" / "sc0 sc1" ...
  const std::string leaf_policy_env = fdg::knob("FDG_ISA_LEAF_POLICY") ? std::string(" ") + fdg::knob("FDG_ISA_LEAF_POLICY") : std::string();
  const std::string root_policy = fdg::knob("FDG_ISA_ROOT_POLICY") ? std::string(" ") + fdg::knob("FDG_ISA_ROOT_POLICY") : std::string(E.streaming || (cs && cs->pooled) ? " nt" : "");
  // Streaming variant: a leaf's last load of the tile and the root stores are non-temporal -- the lines are not needed again, and
  // a read stream with a few stores in it runs 5-10 % faster that way (tools/ubench/tile_ahead.hip: 5.73 -> 6.32 TB/s).  Only for
  // batches whose tiles are whole cache lines (the runtime checks strides and bases): a line shared by two tiles would be
  // fetched twice.  Earlier loads of a leaf that is loaded again stay as they are (the re-load may still find the line in L2).
  // Round 6, the long programs only (Emit::nt_dist >= 0): a load whose leaf IS loaded again in this tile is non-temporal too when that next load
  // comes more than nt_dist (512) leaf loads later -- by then the XCD's 4 MB of L2, which its 128 resident waves share, has been turned over and
  // the line would be fetched from memory anyway, while keeping it costs the stream what every retained line costs (gv_ver4_4's one-wave kernel,
  // leaves re-loaded 2.2 x: no load non-temporal 4.26 ms, last loads only 3.97, distance 1024: 3.46, 512: 3.33, 128: 3.38-3.44, 48: 3.52, every load 3.54).  The graphs with a
  // streaming variant of their own re-load 1.3 x, soon after the eviction, and lose 1-4 % by the same rule (parquet_ver4_4 2.83 -> 2.87 / 2.92 at
  // 128 / 48, gv_sigma5 1.24 -> 1.27 / 1.29): last loads only for them.  profiles/r06_log_nt_sweep.txt, r06_log_sweep_c.txt, r06_log_sweep_d.txt.
  std::vector<uint8_t> final_load(prog.ops.size(), 0);
  if (E.streaming) {
    const long nt_dist = fdg::knob("FDG_ISA_NT_DIST") ? std::atol(fdg::knob("FDG_ISA_NT_DIST")) : E.nt_dist;
    std::vector<long> next_at(p.L + 1, -1);     // ordinal (among the tile's leaf loads, counted from the end) of the leaf's next load
    long ord = 0;
    for (size_t i = prog.ops.size(); i-- > 0;)
      if ((prog.ops[i].kind == M_LD_LEAF || prog.ops[i].kind == M_LD_LEAF_ACC) && prog.ops[i].a < next_at.size()) {
        const uint32_t a = prog.ops[i].a;
        if (next_at[a] < 0 || (nt_dist >= 0 && ord - next_at[a] > nt_dist)) final_load[i] = 1;
        next_at[a] = ord++;
      }
  }
  std::string leaf_policy = leaf_policy_env;
  const std::string DSR = W == 2 ? "ds_read_b128 " : "ds_read_b64 ";
  const std::string DSW = W == 2 ? "ds_write_b128 " : "ds_write_b64 ";

  const char *dbg = fdg::knob("FDG_ISA_DEBUG");
  const bool dbg_noleaf = dbg && std::strstr(dbg, "noleaf");
  const bool dbg_nolds = dbg && std::strstr(dbg, "nolds");
  const bool use_ldexp = fdg::knob("FDG_ISA_NO_LDEXP") == nullptr;
  const bool dbg_novalu = dbg && std::strstr(dbg, "novalu");     // the memory stream of the program alone (waits included)
  const bool dbg_nopanel = dbg && std::strstr(dbg, "nopanel");   // no spill traffic to the HBM panel
  const bool dbg_norecv = dbg && std::strstr(dbg, "norecv");     // pooled / cooperative kernels without their reads of the shared LDS slots
  const bool dbg_nofetch = dbg && std::strstr(dbg, "nopoolfetch");   // ... without the fetches into the pool
  const bool dbg_noacc = dbg && std::strstr(dbg, "noacc");       // no AGPR moves
  const bool dbg_nohead = dbg && std::strstr(dbg, "nohead");     // the loads at the head of a tile count as arrived when the first other operation issues
  bool in_head = true;                                           // (what would hiding the head's memory latency behind the previous tile buy?)
  // experiment (results exact): a wave takes 2^c consecutive tiles, then jumps over the other waves' runs ("chunk<c>", c = 1 .. 9)
  int tile_run = 0;
  if (dbg && std::strstr(dbg, "chunk") && !cs) tile_run = std::max(0, std::min(9, std::atoi(std::strstr(dbg, "chunk") + 5)));
  // experiment (results exact; grids that are a multiple of eight workgroups): workgroup w, which the dispatcher places on XCD w % 8, takes
  // its tiles from the eighth of the batch that belongs to that XCD -- ("xcd") every XCD streams one contiguous region
  const bool tile_xcd = dbg && std::strstr(dbg, "xcd") && !cs && !tile_run;
  // experiment (results exact; grids that are a power of two): in round j workgroup w takes tile j n + ((w + j r) mod n) -- ("rot<r>") the
  // residue class mod 8 of the tiles an XCD reads changes from round to round instead of being w mod 8 for the whole launch.
  // State in the last pair of the constant pool (the experiment is for graphs that leave it free).
  int tile_rot = 0;
  if (dbg && std::strstr(dbg, "rot") && !cs && !tile_run && !tile_xcd) tile_rot = std::max(0, std::atoi(std::strstr(dbg, "rot") + 3));
  const int S_ROTB = S_POOL + 2 * (N_POOL - 1), S_ROTS = S_ROTB + 1;
  E.vm_slack = (dbg && std::strstr(dbg, "noackwait") && !accumulate) ? p.R : 0;
  E.no_vm_wait = dbg && std::strstr(dbg, "novmwait");
  // ---- prologue ------------------------------------------------------------
  if (cs) E.ins("v_and_b32_e32 v0, 63, v0");          // lane within the wave (the workgroup has four waves)
  E.ins("s_load_dwordx8 s[4:11], s[0:1], 0x0");
  E.ins("s_load_dwordx8 s[12:19], s[0:1], 0x20");
  E.ins("s_load_dwordx2 s[20:21], s[0:1], 0x40");
  if (accumulate) E.ins("s_load_dwordx2 " + S2(S_WGT) + ", s[0:1], 0x48");
  const uint32_t n_k = prog.mc_n_k;               // > 0: Monte-Carlo kernel; input columns >= n_k come from the second base
  const bool mc = n_k > 0 || prog.mc_n_t > 0;
  const bool tm_imm = dbg && std::strstr(dbg, "tmimm") && !mc && !rm_bufs && !rl && !cs && W == 1;
  int64_t tm_base = -1;
  const int tile_arg = mc ? 0x80 : 0x50;          // lts, rts follow the other arguments
  E.ins("s_load_dwordx2 " + S2(S_LTS) + ", s[0:1], " + hex32((uint32_t)tile_arg));
  if (!accumulate) E.ins("s_load_dwordx2 " + S2(S_RTS) + ", s[0:1], " + hex32((uint32_t)tile_arg + 8));
  if (mc) {
    E.ins("s_load_dwordx4 s[" + std::to_string(S_LEAF2) + ":" + std::to_string(S_LEAF2 + 3) + "], s[0:1], 0x50");
    E.ins("s_load_dwordx8 s[" + std::to_string(S_PARAM) + ":" + std::to_string(S_PARAM + 7) + "], s[0:1], 0x60");
  }
  E.ins("s_waitcnt lgkmcnt(0)");
  // accumulate mode: R per-lane accumulators, the lane's weight and one temporary live above the value registers
  uint64_t w_seq = 0;
  const uint32_t acc0 = V_BASE + RW * std::max<uint32_t>(prog.n_reg_used, 1);
  // (graphs with 41 ... 124 roots: accumulators in AGPR pairs a[A0 + 2k : +1] behind the program's own; the weight and two temporaries in VGPRs)
  const bool aa = accumulate && prog.params.acc_in_agpr;
  const uint32_t A0 = RW * prog.n_acc_used;
  const uint32_t acc_pairs_v = accumulate ? (aa ? 3 : p.R + 2) : 0;          // VGPR pairs above the values
  const uint32_t wgt0 = aa ? acc0 : acc0 + 2 * p.R;                           // the lane's weight, then the temporary(ies)
  auto vpair = [&](uint32_t b) { return "v[" + std::to_string(b) + ":" + std::to_string(b + 1) + "]"; };
  auto vacc = [&](uint32_t k) { return vpair(acc0 + 2 * k); };                 // (VGPR accumulators only)
  const std::string vwgt = vpair(wgt0), vtmp = vpair(wgt0 + 2), vtmp2 = vpair(wgt0 + 4);
  auto acc_to_tmp2 = [&](uint32_t k) {       // vtmp2 = accumulator k
    E.ins("v_accvgpr_read_b32 v" + std::to_string(wgt0 + 4) + ", a" + std::to_string(A0 + 2 * k));
    E.ins("v_accvgpr_read_b32 v" + std::to_string(wgt0 + 5) + ", a" + std::to_string(A0 + 2 * k + 1));
  };
  auto tmp2_to_acc = [&](uint32_t k) {
    E.ins("v_accvgpr_write_b32 a" + std::to_string(A0 + 2 * k) + ", v" + std::to_string(wgt0 + 4));
    E.ins("v_accvgpr_write_b32 a" + std::to_string(A0 + 2 * k + 1) + ", v" + std::to_string(wgt0 + 5));
  };
  // programs with leaf formulas: two more temporaries (register pairs) above those
  bool has_macro = false;
  uint32_t n_tmp_pairs = 0;
  for (const MOp &o : prog.ops) { if (mop_is_macro(o.kind)) has_macro = true; n_tmp_pairs = std::max(n_tmp_pairs, mop_tmp_pairs(o.kind)); }
  const uint32_t tmp0 = acc0 + 2 * acc_pairs_v;
  auto tpair = [&](uint32_t k) { return "v[" + std::to_string(tmp0 + 2 * k) + ":" + std::to_string(tmp0 + 2 * k + 1) + "]"; };
  const std::string tA = tpair(0), tB = tpair(1), tC = tpair(2), tD = tpair(3);
  auto tAd = [&](int h) { return "v" + std::to_string(tmp0 + h); };
  if (accumulate)
    for (uint32_t k = 0; k < p.R; ++k) {
      if (aa) {
        E.ins("v_accvgpr_write_b32 a" + std::to_string(A0 + 2 * k) + ", 0");
        E.ins("v_accvgpr_write_b32 a" + std::to_string(A0 + 2 * k + 1) + ", 0");
        continue;
      }
      E.ins("v_mov_b32_e32 v" + std::to_string(acc0 + 2 * k) + ", 0");
      E.ins("v_mov_b32_e32 v" + std::to_string(acc0 + 2 * k + 1) + ", 0");
    }
  E.ins("v_lshlrev_b32_e32 " + V(V_LANE8) + ", " + std::to_string(W == 2 ? 4 : 3) + ", v0");      // lane * 8W
  E.ins("v_mul_lo_u32 " + V(V_LEAFOFF) + ", v0, " + S(S_SS));                                    // lane*ss (low 32 bits)
  E.ins("v_lshlrev_b32_e32 " + V(V_LEAFOFF) + ", " + std::to_string(W == 2 ? 4 : 3) + ", " + V(V_LEAFOFF));
  E.ins("v_mul_lo_u32 " + V(V_ROOTOFF) + ", v0, " + S(S_RS));
  E.ins("v_lshlrev_b32_e32 " + V(V_ROOTOFF) + ", " + std::to_string(W == 2 ? 4 : 3) + ", " + V(V_ROOTOFF));
  E.ins("s_lshl_b64 " + S2(S_LS8) + ", " + S2(S_LS) + ", 3");
  E.ins("s_lshl_b64 " + S2(S_RK8) + ", " + S2(S_RK) + ", 3");
  if (mc) E.ins("s_lshl_b64 " + S2(S_LS82) + ", " + S2(S_LS82) + ", 3");
  // row-major variant: per-lane source offset of the LDS-direct loads, and the eight swizzled read addresses
  const uint32_t rm0 = tmp0 + 2 * n_tmp_pairs;                   // v[rm0] = source offset, v[rm0 + 1 + j] = read address of piece j
  const int S_ROW8 = S_DELTA, S_FA = S_DELTA + 2;                // (the delta table is not used by this variant)
  // cooperative section: v[rm0] = lane * 8 + base of this wave's private LDS slots, v[rm0 + 1] = lane * 8 + 64 KB (shared slots
  // 128 and up; the first 128 are reached from V_LANE8 with the 16-bit offset field)
  if (cs) {
    E.ins("v_add_u32_e32 v" + std::to_string(rm0) + ", " + hex32(cs->priv_base_bytes) + ", " + V(V_LANE8));
    E.ins("v_add_u32_e32 v" + std::to_string(rm0 + 1) + ", 0x10000, " + V(V_LANE8));
    E.ins("v_lshlrev_b32_e32 v" + std::to_string(rm0 + 2) + ", 4, v0");          // lane * 16: source offset of a pool fetch (lanes 0..31 carry a leaf's 64 samples)
    if (cs->pooled) {      // (pooled programs load no leaf into a register: the delta table's registers are free)
      E.ins("s_mov_b32 " + S(S_DELTA + 4) + ", 0");                               // lanes 32..63: the second leaf of a paired fetch (v[rm0 + 3]: its lanes' offsets)
      E.ins("s_mov_b32 " + S(S_DELTA + 5) + ", -1");       // (not 0xffffffff: the assembler makes that the inline constant -1 too, but isa_size would count a literal)
    }
    if (cs->slack) {
      // Progress words (round 6): wave w owns the 256 bytes at flag_base + 256 w -- every lane writes its own dword, so the store has no bank
      // conflict -- and publishes there the number of sync points it has reached (counted over the whole launch, from 1024).  A reader's lane l
      // looks at wave l % n_wave's first dword: one ds_read_b32 brings every wave's progress, one v_cmp tells whether any of them is behind.
      E.ins("v_lshlrev_b32_e32 v" + std::to_string(rm0 + 4) + ", 2, v0");
      E.ins("v_add_u32_e32 v" + std::to_string(rm0 + 4) + ", " + hex32(cs->flag_base_bytes + 256u * cs->wave) + ", v" + std::to_string(rm0 + 4));      // my word of this lane
      E.ins("v_and_b32_e32 v" + std::to_string(rm0 + 5) + ", " + std::to_string(cs->n_wave - 1) + ", v0");
      E.ins("v_lshlrev_b32_e32 v" + std::to_string(rm0 + 5) + ", 8, v" + std::to_string(rm0 + 5));
      E.ins("v_add_u32_e32 v" + std::to_string(rm0 + 5) + ", " + hex32(cs->flag_base_bytes) + ", v" + std::to_string(rm0 + 5));                          // wave (lane % n_wave)'s word
      E.ins("s_movk_i32 " + S(S_DELTA + 6) + ", 0x400");                          // S_PROG: sync points reached
      E.ins("v_mov_b32_e32 " + V(V_TMP) + ", " + S(S_DELTA + 6));
      E.ins("ds_write_b32 v" + std::to_string(rm0 + 4) + ", " + V(V_TMP));
      E.ins("s_waitcnt lgkmcnt(0)");
      E.ins("s_barrier");                                                         // every wave's word is initialised before anyone looks
      E.ins("ds_read_b32 v" + std::to_string(rm0 + 6) + ", v" + std::to_string(rm0 + 5));
      E.ins("s_waitcnt lgkmcnt(0)");
    }
  }
  if (rl) {
    E.ins("v_mul_u32_u24_e32 v" + std::to_string(rm0) + ", " + hex32(8u * p.L) + ", v0");      // lane * 8 L: this lane's row inside the image
    E.ins("v_lshlrev_b32_e32 v" + std::to_string(rm0 + 1) + ", 4, v0");                        // lane * 16: source offset of the linear loads
  }
  if (rm_bufs) {
    // v[rm0] / v[rm0 + 9]: source offsets of the LDS-direct loads of even / odd instructions n of a chunk -- lane l fetches
    // piece (l % 8) ^ s of row 8 n + l / 8, s = (row / 2) % 8 = (4 n + l / 16) % 8: l / 16 for even n, that ^ 4 for odd n;
    // v[rm0 + 1 + j]: where lane = row r finds piece j: r * 128 + ((j ^ ((r / 2) % 8)) * 16).  With this swizzle the 32 rows
    // a ds_read_b64 services per LDS cycle pair fall on 16 distinct bank groups x 2 halves (two-way instead of four-way
    // conflict with s = r % 8), and the 16 rows of a ds_read_b128 group on all 64 banks.
    const std::string vg = "v" + std::to_string(rm0), vg2 = "v" + std::to_string(rm0 + 9);
    auto vj = [&](int j) { return "v" + std::to_string(rm0 + 1 + j); };
    E.ins("v_lshrrev_b32_e32 " + vj(0) + ", 3, v0");                                   // l / 8
    E.ins("v_mul_lo_u32 " + vj(0) + ", " + vj(0) + ", " + S(S_SS));                    // (l / 8) rows further
    E.ins("v_lshlrev_b32_e32 " + vj(0) + ", 3, " + vj(0));
    E.ins("v_lshrrev_b32_e32 " + vj(1) + ", 4, v0");                                   // l / 16
    E.ins("v_and_b32_e32 " + vj(2) + ", 7, v0");                                       // l % 8
    E.ins("v_xor_b32_e32 " + vj(1) + ", " + vj(1) + ", " + vj(2));                     // piece this lane fetches (even n)
    E.ins("v_xor_b32_e32 " + vj(2) + ", 4, " + vj(1));                                 // ... (odd n)
    E.ins("v_lshlrev_b32_e32 " + vj(1) + ", 4, " + vj(1));
    E.ins("v_lshlrev_b32_e32 " + vj(2) + ", 4, " + vj(2));
    E.ins("v_add_u32_e32 " + vg + ", " + vj(0) + ", " + vj(1));
    E.ins("v_add_u32_e32 " + vg2 + ", " + vj(0) + ", " + vj(2));
    E.ins("v_lshlrev_b32_e32 " + V(V_TMP) + ", 7, v0");                                // row * 128
    E.ins("v_bfe_u32 " + V(V_TMP + 1) + ", v0, 1, 3");                                 // (row / 2) % 8
    for (int j = 0; j < 8; ++j) {
      E.ins("v_xor_b32_e32 " + vj(j) + ", " + std::to_string(j) + ", " + V(V_TMP + 1));
      E.ins("v_lshlrev_b32_e32 " + vj(j) + ", 4, " + vj(j));
      E.ins("v_add_u32_e32 " + vj(j) + ", " + vj(j) + ", " + V(V_TMP));
    }
    E.ins("s_lshl_b64 " + S2(S_ROW8) + ", " + S2(S_SS) + ", 6");                       // eight rows in bytes
  }
  // Leaf addresses: consecutive loads mostly step by +1 leaf (leaves are numbered in first-visit order and
  // the schedule visits them nearly in that order), so the column pointer is advanced by an add of the
  // stride (2 scalar ops) instead of being rebuilt from the leaf index (6); the next most frequent positive
  // steps get their stride multiple precomputed once per wave.
  std::vector<int64_t> delta_tab;
  {
    std::map<int64_t, int> hist;
    int64_t last = -1;
    for (const MOp &o : prog.ops) if (o.kind == M_LD_LEAF || o.kind == M_LD_LEAF_ACC) {
      const bool same_space = !mc || (o.a < n_k && last < (int64_t)n_k);     // only steps inside the first column space use the table
      if (last >= 0 && same_space && (int64_t)o.a - last > 1) hist[(int64_t)o.a - last]++;
      last = o.a;
    }
    std::vector<std::pair<int, int64_t>> v;
    for (auto &kv : hist) if (kv.second >= 2) v.push_back({kv.second, kv.first});
    std::sort(v.begin(), v.end(), [](const auto &x, const auto &y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
    for (size_t i = 0; i < v.size() && i < (size_t)N_DELTA && !rm_bufs && !rl; ++i) delta_tab.push_back(v[i].second);
  }
  for (size_t k = 0; k < delta_tab.size(); ++k) {
    const int d = S_DELTA + 2 * (int)k;
    const std::string ks = hex32((uint32_t)delta_tab[k]);
    E.ins("s_mul_i32 " + S(d) + ", " + S(S_LS8) + ", " + ks);
    E.ins("s_mul_hi_u32 " + S(d + 1) + ", " + S(S_LS8) + ", " + ks);
    E.ins("s_mul_i32 " + S(S_X) + ", " + S(S_LS8 + 1) + ", " + ks);
    E.ins("s_add_u32 " + S(d + 1) + ", " + S(d + 1) + ", " + S(S_X));
  }
  // Edge factors that have no inline encoding: a graph uses a handful of distinct ones (spin / symmetry
  // factors), so they live in SGPR pairs for the whole kernel instead of being moved in before every use.
  std::vector<uint64_t> pool;
  {
    std::map<uint64_t, int> hist;
    auto count = [&](double f) { bool inl; f64_inline(f, inl); if (!inl) { uint64_t u; std::memcpy(&u, &f, 8); hist[u]++; } };
    for (const MOp &o : prog.ops) {
      if ((o.kind == M_MULC || o.kind == M_FMAC || o.kind == M_ADDC || o.kind == M_FIXZ || o.kind == M_SELC || o.kind == M_FMAK) && !o.param) count(o.imm);
      if (o.kind == M_EXP) { count(kLog2e); count(-kLn2Hi); count(-kLn2Lo); for (int k = 0; k <= kExpDeg; ++k) count(kExpC[k]); }
    }
    std::vector<std::pair<int, uint64_t>> v;
    for (auto &kv : hist) v.push_back({kv.second, kv.first});
    std::sort(v.begin(), v.end(), [](const auto &x, const auto &y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
    for (size_t i = 0; i < v.size() && i < (size_t)(mc ? N_POOL_MC : (has_macro ? N_POOL : 16)); ++i) pool.push_back(v[i].second);
  }
  for (size_t k = 0; k < pool.size(); ++k) {
    E.ins("s_mov_b32 " + S(S_POOL + 2 * (int)k) + ", " + hex32((uint32_t)pool[k]));
    E.ins("s_mov_b32 " + S(S_POOL + 2 * (int)k + 1) + ", " + hex32((uint32_t)(pool[k] >> 32)));
  }
  // ntiles = ceil(B / tile)   (W = 2 is only launched on a multiple of 128 samples)
  E.ins("s_add_u32 " + S(S_X) + ", " + S(S_B) + ", " + std::to_string((1 << TSH) - 1));
  E.ins("s_addc_u32 " + S(S_X + 1) + ", " + S(S_B + 1) + ", 0");
  E.ins("s_lshr_b64 " + S2(S_X) + ", " + S2(S_X) + ", " + std::to_string(TSH));
  E.ins("s_mov_b32 " + S(S_NTILES) + ", " + S(S_X));
  if (tile_run) E.ins("s_lshl_b32 " + S(S_TILE) + ", s2, " + std::to_string(tile_run));
  else if (tile_xcd) {
    E.ins("s_and_b32 " + S(S_X) + ", s2, 7");                                      // XCD of this workgroup
    E.ins("s_add_u32 " + S(S_T) + ", " + S(S_NTILES) + ", 7");
    E.ins("s_lshr_b32 " + S(S_T) + ", " + S(S_T) + ", 3");                         // tiles per XCD
    E.ins("s_mul_i32 " + S(S_A) + ", " + S(S_X) + ", " + S(S_T));                  // first tile of the XCD's range
    E.ins("s_lshr_b32 " + S(S_A + 1) + ", s2, 3");                                 // this workgroup's index inside its XCD
    E.ins("s_add_u32 " + S(S_TILE) + ", " + S(S_A) + ", " + S(S_A + 1));
    E.ins("s_add_u32 " + S(S_A) + ", " + S(S_A) + ", " + S(S_T));
    E.ins("s_min_u32 " + S(S_NTILES) + ", " + S(S_A) + ", " + S(S_NTILES));        // end of the range (the loop's bound from here on)
  }
  else {
    E.ins("s_mov_b32 " + S(S_TILE) + ", s2");
    if (tile_rot) { E.ins("s_mov_b32 " + S(S_ROTB) + ", 0"); E.ins("s_mov_b32 " + S(S_ROTS) + ", s2"); }
  }
  E.ins("s_mul_i32 " + S(S_X) + ", s2, " + hex32(cs ? cs->panel_wg_bytes : panel_bytes_per_wave));
  E.ins("s_mul_hi_u32 " + S(S_X + 1) + ", s2, " + hex32(cs ? cs->panel_wg_bytes : panel_bytes_per_wave));
  E.ins("s_add_u32 " + S(S_PANEL) + ", " + S(S_WS) + ", " + S(S_X));
  E.ins("s_addc_u32 " + S(S_PANEL + 1) + ", " + S(S_WS + 1) + ", " + S(S_X + 1));
  if (cs && cs->panel_prefix_bytes) {
    E.ins("s_add_u32 " + S(S_PANEL) + ", " + S(S_PANEL) + ", " + hex32(cs->panel_prefix_bytes));
    E.ins("s_addc_u32 " + S(S_PANEL + 1) + ", " + S(S_PANEL + 1) + ", 0");
  }
  // Panel addresses: slot s lives at panel + SLOT s, and the instruction's signed 13-bit offset reaches 4 KB either way
  // of an SGPR base.  The pool's unused pairs hold panel + 8192 k for the 8 KB windows the program touches most, set
  // once per wave, so that those accesses need no scalar add (the others build their base in S_A as before).
  std::map<uint32_t, int> panel_base;          // window k (bytes 8192 k - 4096 .. 8192 k + 4095) -> first SGPR of its base pair
  if (!mc) {
    std::map<uint32_t, uint64_t> hist;
    for (const MOp &o : prog.ops) {
      if (o.kind != M_LD_MEM && o.kind != M_ST_MEM) continue;
      const uint32_t slot = o.kind == M_LD_MEM ? o.a : o.d;
      const uint64_t byte = (uint64_t)slot * SLOT;
      if (byte >= 4096) hist[(uint32_t)((byte + 4096) / 8192)]++;
    }
    std::vector<std::pair<uint64_t, uint32_t>> v;
    for (auto &kv : hist) v.push_back({kv.second, kv.first});
    std::sort(v.begin(), v.end(), [](const auto &x, const auto &y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
    for (size_t i = 0; i < v.size() && pool.size() + i < (size_t)N_POOL; ++i) {
      const int r = S_POOL + 2 * (int)(pool.size() + i);
      panel_base[v[i].second] = r;
      E.ins("s_add_u32 " + S(r) + ", " + S(S_PANEL) + ", " + hex32(v[i].second * 8192u));
      E.ins("s_addc_u32 " + S(r + 1) + ", " + S(S_PANEL + 1) + ", 0");
    }
  }
  E.ins("s_cmp_ge_u32 " + S(S_TILE) + ", " + S(S_NTILES));
  E.ins("s_cbranch_scc0 .Ltile" + sfx);
  E.ins("s_endpgm");   // (the host never launches more waves than tiles)
  os << ".Ltile" << sfx << ":\n";
  if (cs && cs->slack) E.ins("s_mov_b32 " + S(S_DELTA + 7) + ", " + S(S_DELTA + 6));      // S_G0: the progress count at the start of this tile
  E.ins("s_mov_b32 " + S(S_X + 1) + ", 0");
  E.ins("s_mov_b32 " + S(S_X) + ", " + S(S_TILE));
  E.ins("s_lshl_b64 " + S2(S_X) + ", " + S2(S_X) + ", " + std::to_string(TSH));                 // b0
  if (W == 1) {
    E.ins("s_sub_u32 " + S(S_T) + ", " + S(S_B) + ", " + S(S_X));           // B - b0 (low); high decides >= 64
    E.ins("s_subb_u32 " + S(S_T + 1) + ", " + S(S_B + 1) + ", " + S(S_X + 1));
    E.ins("s_cmp_lg_u32 " + S(S_T + 1) + ", 0");
    E.ins("s_cselect_b32 " + S(S_T) + ", 64, " + S(S_T));
    E.ins("s_min_u32 " + S(S_T) + ", " + S(S_T) + ", 64");
    E.ins("s_bfm_b64 " + S2(S_A) + ", " + S(S_T) + ", 0");
    E.ins("s_cmp_ge_u32 " + S(S_T) + ", 64");
    E.ins("s_cselect_b64 exec, -1, " + S2(S_A));
  }
  // dst = base + 8 * tile * stride   (tile: 32 bits, stride: the 64-bit tile stride in elements)
  auto tile_base = [&](int dst, int base, int stride) {
    E.ins("s_mul_i32 " + S(S_A) + ", " + S(S_TILE) + ", " + S(stride));
    E.ins("s_mul_hi_u32 " + S(S_A + 1) + ", " + S(S_TILE) + ", " + S(stride));
    E.ins("s_mul_i32 " + S(S_T) + ", " + S(S_TILE) + ", " + S(stride + 1));
    E.ins("s_add_u32 " + S(S_A + 1) + ", " + S(S_A + 1) + ", " + S(S_T));
    E.ins("s_lshl_b64 " + S2(S_A) + ", " + S2(S_A) + ", 3");
    E.ins("s_add_u32 " + S(dst) + ", " + S(base) + ", " + S(S_A));
    E.ins("s_addc_u32 " + S(dst + 1) + ", " + S(base + 1) + ", " + S(S_A + 1));
  };
  tile_base(S_LT, S_LEAF, S_LTS);
  if (mc) tile_base(S_LT2, S_LEAF2, S_LTS);
  if (!accumulate) tile_base(S_RT, S_ROOT, S_RTS);
  if (accumulate) {
    // w = weight ? weight[b0 + lane] : 1.0   (consumed at the first root, long after this load)
    E.ins("v_mov_b32_e32 v" + std::to_string(wgt0) + ", 0");
    E.ins("v_mov_b32_e32 v" + std::to_string(wgt0 + 1) + ", 0x3ff00000");
    E.ins("s_cmp_eq_u64 " + S2(S_WGT) + ", 0");
    E.ins("s_cbranch_scc1 .Lnow" + sfx);
    E.ins("s_lshl_b64 " + S2(S_A) + ", " + S2(S_X) + ", 3");
    E.ins("s_add_u32 " + S(S_A) + ", " + S(S_A) + ", " + S(S_WGT));
    E.ins("s_addc_u32 " + S(S_A + 1) + ", " + S(S_A + 1) + ", " + S(S_WGT + 1));
    E.ins("global_load_dwordx2 " + vwgt + ", " + V(V_LANE8) + ", " + S2(S_A));
    os << ".Lnow" << sfx << ":\n";
    w_seq = ++E.vm_issued;     // counted on both paths: a phantom older op only makes later waits conservative
  }

  uint64_t rl_ready = 0;
  if (rl) {
    // the tile's block: 512 L bytes from S_LT on, 1 KB per instruction (the last one half a wave wide when L is odd); the image's previous
    // readers have drained (lgkmcnt(0) at the end of the tile)
    const uint32_t n_inst = (512u * p.L + 1023u) / 1024u;
    E.ins("s_mov_b64 " + S2(S_DELTA + 2) + ", " + S2(S_LT));
    for (uint32_t n = 0; n < n_inst; ++n) {
      if (n && n % 4 == 0) {
        E.ins("s_add_u32 " + S(S_DELTA + 2) + ", " + S(S_DELTA + 2) + ", 0x1000");
        E.ins("s_addc_u32 " + S(S_DELTA + 3) + ", " + S(S_DELTA + 3) + ", 0");
      }
      const bool half = (n + 1) * 1024u > 512u * p.L;
      if (half) E.ins("s_mov_b64 exec, 0xffffffff");
      // (the instruction's offset moves the LDS address as well as the global one: M0 only changes every four instructions)
      if (n % 4 == 0 || half) E.ins("s_mov_b32 m0, " + hex32(stage_base + (n / 4) * 4096u));
      E.ins("global_load_lds_dwordx4 v" + std::to_string(rm0 + 1) + ", " + S2(S_DELTA + 2) + " offset:" + std::to_string((n % 4) * 1024u) + " nt");
      if (half) E.ins("s_mov_b64 exec, -1");
      ++E.vm_issued;
    }
    rl_ready = E.vm_issued;
  }
  auto panel_operand = [&](uint32_t slot) -> std::string {
    const uint64_t byte = (uint64_t)slot * SLOT;
    const uint64_t hi = byte & ~4095ull, lo = byte & 4095ull;
    if (hi == 0) return S2(S_PANEL) + " offset:" + std::to_string(lo);
    const auto it = panel_base.find((uint32_t)((byte + 4096) / 8192));
    if (it != panel_base.end()) return S2(it->second) + " offset:" + std::to_string((int64_t)byte - (int64_t)it->first * 8192);
    E.ins("s_add_u32 " + S(S_A) + ", " + S(S_PANEL) + ", " + hex32((uint32_t)hi));
    E.ins("s_addc_u32 " + S(S_A + 1) + ", " + S(S_PANEL + 1) + ", " + hex32((uint32_t)(hi >> 32)));
    return S2(S_A) + " offset:" + std::to_string(lo);
  };
  auto valu2 = [&](const std::string &opc, const MOp &o, const std::string &second_lo, const std::string &second_hi) {
    E.ins(opc + vlo(o.d) + ", " + (o.nega ? "-" : "") + vlo(o.a) + ", " + second_lo);
    if (W == 2) E.ins(opc + vhi(o.d) + ", " + (o.nega ? "-" : "") + vhi(o.a) + ", " + second_hi);
  };

  // a constant multiplier as an instruction operand: inline encoding, resident SGPR pair, or moved in just now
  auto const_operand = [&](double imm) -> std::string {
    bool inl;
    std::string c = f64_inline(imm, inl);
    if (inl) return c;
    uint64_t u;
    std::memcpy(&u, &imm, 8);
    for (size_t k = 0; k < pool.size(); ++k) if (pool[k] == u) return S2(S_POOL + 2 * (int)k);
    E.ins("s_mov_b32 " + S(S_C) + ", " + hex32((uint32_t)u));
    E.ins("s_mov_b32 " + S(S_C + 1) + ", " + hex32((uint32_t)(u >> 32)));
    return S2(S_C);
  };
  // the same constant as an SGPR pair (first register), for v_mov_b32 of its halves
  auto const_sgpr = [&](double imm) -> int {
    uint64_t u;
    std::memcpy(&u, &imm, 8);
    for (size_t k = 0; k < pool.size(); ++k) if (pool[k] == u) return S_POOL + 2 * (int)k;
    E.ins("s_mov_b32 " + S(S_C) + ", " + hex32((uint32_t)u));
    E.ins("s_mov_b32 " + S(S_C + 1) + ", " + hex32((uint32_t)(u >> 32)));
    return S_C;
  };
  auto vd = [&](uint32_t r, int h) { return "v" + std::to_string(V_BASE + RW * r + h); };   // one dword of a value register
  // the constant of a micro-op: a kernel argument when tagged, else as above
  auto op_const = [&](const MOp &o) -> std::string { return o.param ? S2(S_PARAM + 2 * (o.param - 1)) : const_operand(o.imm); };
  auto op_const_sgpr = [&](const MOp &o) -> int { return o.param ? S_PARAM + 2 * (o.param - 1) : const_sgpr(o.imm); };
  // ---- row-major variant: which chunk sits in which staging buffer when (rm_plan above) ---------------------------
  const uint32_t rm_full = p.L / RM_CHUNK;
  auto rm_chunk_of = [&](uint32_t leaf) { return leaf < rm_full * RM_CHUNK ? leaf / RM_CHUNK : rm_full; };
  auto rm_chunk_start = [&](uint32_t c) { return c < rm_full ? c * RM_CHUNK : p.L - RM_CHUNK; };
  std::vector<std::vector<RmFetch>> rm_fetch;       // [op index] fetches issued in front of that op
  std::vector<int> rm_ld_buf;                       // [op index] staging buffer an LD_LEAF reads, -1 = gathered from memory
  if (rm_bufs) rm_plan(p, prog, rm_bufs, rm_fetch, rm_ld_buf);
  // Cache policy of the LDS-direct loads (experiment knob FDG_ISA_RM_POLICY="nt" ...).  Non-temporal loads stream 6.9 instead of
  // 6.1 TB/s in a bare loop over rows that are whole cache lines (tools/ubench/rm_stream.hip), but a row of L doubles is not: the
  // 128-byte segments of consecutive chunks share lines, and a line fetched non-temporally is fetched again from memory for the
  // next chunk (measured: parquet_sigma4 6.2 -> 3.6e9 evals/s).  Plain loads it is.
  // Cache policy of the pooled kernels' fetches: NON-TEMPORAL (round 6).  A fetch lands in LDS and is read from there; its line is of no use to
  // the caches, and a plain LDS-direct load costs the CU far more than a streaming one: gv_ver4_4 3.65 -> 3.06 ms per 524 288 samples, bit for
  // bit the same (profiles/r06_log_pool_sweep.txt; "sc1" / "sc0 sc1" change nothing, "sc0 sc1 nt" = "nt").  The fetches cost the same with one
  // or two waves per SIMD and whichever waves issue them -- a property of the CU's path to memory, not of the issuing wave.
  // FDG_POOL_FETCH_POLICY="plain" / "sc1" / ... overrides.
  const std::string panel_policy = fdg::knob("FDG_ISA_PANEL_POLICY") && fdg::knob("FDG_ISA_PANEL_POLICY")[0] ? std::string(" ") + fdg::knob("FDG_ISA_PANEL_POLICY") : std::string();   // (experiment)
  const char *pool_policy_env = fdg::knob("FDG_POOL_FETCH_POLICY");
  const std::string pool_policy = !pool_policy_env ? std::string(" nt") : (std::string(pool_policy_env) == "plain" || !pool_policy_env[0] ? std::string() : std::string(" ") + pool_policy_env);
  const std::string rm_policy = fdg::knob("FDG_ISA_RM_POLICY") && fdg::knob("FDG_ISA_RM_POLICY")[0] ? std::string(" ") + fdg::knob("FDG_ISA_RM_POLICY") : std::string();
  std::vector<uint64_t> rm_ready(rm_bufs, 0);        // vm sequence number of the last load of the chunk in each buffer
  auto rm_emit_fetch = [&](const RmFetch &f) {
    // the buffer's previous readers have been issued; their data must have left the LDS before it is overwritten
    if (E.lg_done < E.lg_issued) { E.ins("s_waitcnt lgkmcnt(0)"); E.lg_done = E.lg_issued; for (auto &pp : E.pend) if (pp.first == 2) pp.first = 0; }
    E.ins("s_mov_b32 m0, " + hex32(stage_base + f.buf * RM_BUF_BYTES));
    const uint64_t off = (uint64_t)rm_chunk_start(f.chunk) * 8;
    E.ins("s_add_u32 " + S(S_FA) + ", " + S(S_LT) + ", " + hex32((uint32_t)off));
    E.ins("s_addc_u32 " + S(S_FA + 1) + ", " + S(S_LT + 1) + ", 0");
    for (uint32_t n = 0; n < 8; ++n) {
      if (n) {   // (two scalar ops between the write of M0 and the load that reads it)
        E.ins("s_mov_b32 m0, " + hex32(stage_base + f.buf * RM_BUF_BYTES + n * 1024));
        E.ins("s_add_u32 " + S(S_FA) + ", " + S(S_FA) + ", " + S(S_ROW8));
        E.ins("s_addc_u32 " + S(S_FA + 1) + ", " + S(S_FA + 1) + ", " + S(S_ROW8 + 1));
      }
      E.ins("global_load_lds_dwordx4 v" + std::to_string((n & 1) ? rm0 + 9 : rm0) + ", " + S2(S_FA) + rm_policy);
      ++E.vm_issued;
    }
    rm_ready[f.buf] = E.vm_issued;
  };

  // ---- body ------------------------------------------------------------------
  int64_t last_leaf = -1;
  // One s_waitcnt can serve several consumers: when an op has to wait for a load, the wait also covers what
  // the next few ops need of the loads issued so far (they complete in order and were issued long before,
  // so this costs nothing and saves issue slots).
  const char *wl_env = fdg::knob("FDG_ISA_WAIT_LOOKAHEAD");
  const size_t wait_look = wl_env ? (size_t)std::max(0, std::atoi(wl_env)) : 6;
  auto vm_seq_needed = [&](const MOp &q) -> uint64_t {
    uint64_t sq = 0;
    auto use = [&](uint32_t r) { if (r < E.pend.size() && E.pend[r].first == 1) sq = std::max(sq, E.pend[r].second); };
    switch (q.kind) {
      case M_LD_ACC: use(q.d); if (q.a < E.pend_acc.size()) sq = std::max(sq, E.pend_acc[q.a]); break;
      case M_LD_LEAF: case M_LD_LDS: case M_LD_MEM: case M_RECV: use(q.d); break;
      case M_ST_LDS: case M_ST_MEM: case M_ST_ACC: case M_ROOT: case M_SEND: use(q.a); break;
      case M_MUL: case M_ADD: use(q.a); use(q.b); use(q.d); break;
      case M_FMA: use(q.a); use(q.b); use(q.c); use(q.d); break;
      case M_FMAC: use(q.a); use(q.c); use(q.d); break;
      case M_MULC: case M_MOV: case M_ADDC: case M_EXP: case M_RCP: case M_FIXZ: case M_SELC: case M_DIV1: use(q.a); use(q.d); break;
      case M_CONST: use(q.d); break;
      case M_FMAK: use(q.a); use(q.b); use(q.d); break;
      case M_SEL: use(q.a); use(q.b); use(q.c); use(q.d); break;
      default: break;
    }
    return sq;
  };
  size_t op_index = 0;
  std::vector<std::pair<uint64_t, uint32_t>> pool_pending;      // (vm sequence number, first epoch in which it is read) of this wave's pool fetches in flight
  uint32_t bar_seen = 0;
  uint64_t flag_seq = 0;      // lgkm sequence number of the pending read of the progress words (flag synchronisation)
  bool pool_exec_low = false;
  for (const MOp &o : prog.ops) {
    if (rm_bufs) for (const RmFetch &f : rm_fetch[op_index]) rm_emit_fetch(f);
    const size_t this_op = op_index;
    if (in_head && o.kind != M_LD_LEAF && o.kind != M_LD_LEAF_ACC) {
      in_head = false;
      if (dbg_nohead) { E.vm_done = E.vm_issued; for (auto &pp : E.pend) if (pp.first == 1) pp.first = 0; for (auto &pa : E.pend_acc) pa = 0; }
    }
    {
      const size_t i = op_index++;
      uint64_t need = vm_seq_needed(o);
      if (need > E.vm_done && wait_look) {
        for (size_t j = i + 1; j < prog.ops.size() && j <= i + wait_look; ++j) need = std::max(need, vm_seq_needed(prog.ops[j]));
        E.wait_vm(need);
      }
    }
    switch (o.kind) {
      case M_LD_LEAF_ACC:
      case M_LD_LEAF:
        if (dbg_noleaf) break;   // timing experiments only (results are garbage)
        if (o.kind == M_LD_LEAF) E.wait_reg(o.d);
        if (E.streaming && leaf_policy_env.empty()) leaf_policy = final_load[this_op] ? " nt" : "";
        if (rl) {              // lane = row: image[row * 8 L + 8 leaf]
          E.wait_vm(rl_ready);
          E.ins("ds_read_b64 " + vall(o.d) + ", v" + std::to_string(rm0) + " offset:" + std::to_string(stage_base + o.a * 8u));
          E.pend[o.d] = {2, ++E.lg_issued};
          break;
        }
        if (rm_bufs) {
          const int b = rm_ld_buf[this_op];
          if (b >= 0) {          // from the staging buffer: lane = row, piece (leaf - chunk start) / 2, half (leaf - chunk start) % 2
            const uint32_t c = o.a - rm_chunk_start(rm_chunk_of(o.a));
            E.wait_vm(rm_ready[(size_t)b]);
            E.ins("ds_read_b64 " + vall(o.d) + ", v" + std::to_string(rm0 + 1 + c / 2) + " offset:" +
                  std::to_string(stage_base + (uint32_t)b * RM_BUF_BYTES + (c % 2) * 8));
            E.pend[o.d] = {2, ++E.lg_issued};
          } else {               // gathered: eight bytes of each lane's own row
            E.ins("s_add_u32 " + S(S_LP) + ", " + S(S_LT) + ", " + hex32(o.a * 8u));
            E.ins("s_addc_u32 " + S(S_LP + 1) + ", " + S(S_LT + 1) + ", 0");
            E.ins(LD + vall(o.d) + ", " + V(V_LEAFOFF) + ", " + S2(S_LP) + leaf_policy);
            E.pend[o.d] = {1, ++E.vm_issued};
          }
          break;
        }
        if (tm_imm) {          // experiment: leaf stride 64 assumed (tile-major): leaf i at tile base + 512 i, reached by the instruction's 13-bit offset
          const int64_t byte = 512ll * (int64_t)o.a;
          if (tm_base < 0 || byte - tm_base < -4096 || byte - tm_base > 4095) {
            tm_base = byte + 4096;             // this leaf at -4096, the fifteen behind it up to +3584
            E.ins("s_add_u32 " + S(S_LP) + ", " + S(S_LT) + ", " + hex32((uint32_t)tm_base));
            E.ins("s_addc_u32 " + S(S_LP + 1) + ", " + S(S_LT + 1) + ", 0");
          }
          const std::string off = " offset:" + std::to_string(byte - tm_base);
          if (o.kind == M_LD_LEAF_ACC) {
            E.ins(LD + "a[" + std::to_string(RW * o.d) + ":" + std::to_string(RW * o.d + RW - 1) + "], " + V(V_LEAFOFF) + ", " + S2(S_LP) + off + leaf_policy);
            if (E.pend_acc.size() <= o.d) E.pend_acc.resize(o.d + 1, 0);
            E.pend_acc[o.d] = ++E.vm_issued;
            break;
          }
          E.ins(LD + vall(o.d) + ", " + V(V_LEAFOFF) + ", " + S2(S_LP) + off + leaf_policy);
          E.pend[o.d] = {1, ++E.vm_issued};
          break;
        }
        {
          const int64_t step = last_leaf >= 0 ? (int64_t)o.a - last_leaf : 0;
          const bool second = mc && o.a >= n_k;                      // a time: second base, its own column stride
          const bool same_space = !mc || last_leaf < 0 || (second == (last_leaf >= (int64_t)n_k));
          int dreg = -1;
          if (last_leaf >= 0 && same_space && step == 1) dreg = second ? S_LS82 : S_LS8;
          for (size_t k = 0; k < delta_tab.size() && last_leaf >= 0 && same_space && !second; ++k) if (delta_tab[k] == step) dreg = S_DELTA + 2 * (int)k;
          if (last_leaf >= 0 && step == 0) {
            // same column again: pointer already there
          } else if (dreg >= 0) {
            E.ins("s_add_u32 " + S(S_LP) + ", " + S(S_LP) + ", " + S(dreg));
            E.ins("s_addc_u32 " + S(S_LP + 1) + ", " + S(S_LP + 1) + ", " + S(dreg + 1));
          } else if (second) {
            emit_scaled_addr(E, S_LP, S_LT2, S_LS82, o.a - n_k);
          } else {
            emit_scaled_addr(E, S_LP, S_LT, S_LS8, o.a);
          }
          last_leaf = o.a;
        }
        if (o.kind == M_LD_LEAF_ACC) {       // lands in an AGPR pair; its previous occupant was read (v_accvgpr_read) earlier in program order
          E.ins(LD + "a[" + std::to_string(RW * o.d) + ":" + std::to_string(RW * o.d + RW - 1) + "], " + V(V_LEAFOFF) + ", " + S2(S_LP) + leaf_policy);
          if (E.pend_acc.size() <= o.d) E.pend_acc.resize(o.d + 1, 0);
          E.pend_acc[o.d] = ++E.vm_issued;
          break;
        }
        E.ins(LD + vall(o.d) + ", " + V(V_LEAFOFF) + ", " + S2(S_LP) + leaf_policy);
        E.pend[o.d] = {1, ++E.vm_issued};
        break;
      case M_LD_MEM: {
        if (dbg_nopanel) break;
        E.wait_reg(o.d);
        const std::string opnd = panel_operand(o.a);
        E.ins(LD + vall(o.d) + ", " + V(V_LANE8) + ", " + opnd + panel_policy);
        E.pend[o.d] = {1, ++E.vm_issued};
        break;
      }
      case M_ST_MEM: {
        if (dbg_nopanel) break;
        E.wait_reg(o.a);
        const std::string opnd = panel_operand(o.d);
        E.ins(ST + V(V_LANE8) + ", " + vall(o.a) + ", " + opnd + panel_policy);
        ++E.vm_issued;
        break;
      }
      case M_LD_LDS:
        if (dbg_nolds) break;
        E.wait_reg(o.d);
        E.ins(DSR + vall(o.d) + ", " + (cs ? "v" + std::to_string(rm0) : V(V_LANE8)) + " offset:" + std::to_string(o.a * SLOT));
        E.pend[o.d] = {2, ++E.lg_issued};
        break;
      case M_RECV:           // a value another wave published before the last barrier
        if (dbg_norecv) break;
        E.wait_reg(o.d);
        E.ins(DSR + vall(o.d) + ", " + (o.a < 128 ? V(V_LANE8) : "v" + std::to_string(rm0 + 1)) + " offset:" + std::to_string((o.a % 128) * SLOT));
        E.pend[o.d] = {2, ++E.lg_issued};
        break;
      case M_POOL_FETCH: {   // shared[d .. d + b - 1] = leaf[a .. a + b - 1]: 16 bytes per lane straight into the LDS slot(s), no register in between
        if (dbg_nofetch) break;
        const bool pair = o.b == 2;          // 64 lanes: two leaves adjacent in the tile (leaf stride 64, checked at launch) into two adjacent slots
        if (!pair && !pool_exec_low) { E.ins("s_mov_b64 exec, 0xffffffff"); pool_exec_low = true; }
        if (pair && pool_exec_low) { E.ins("s_mov_b64 exec, -1"); pool_exec_low = false; }
        // The tile's leaves lie within 2 GB of its first (checked at launch), so a leaf's place is a 32-bit offset from the tile's base: it goes
        // into the lanes' offset register by ONE vector add (round 6: M0 first -- the wait state it needs before the load is then filled by the
        // address arithmetic --, no 64-bit scalar address: four instructions per fetch instead of seven; the fetch sequences are issue overhead
        // of a lone wave now that the loads themselves are non-temporal).
        E.ins("s_mov_b32 m0, " + hex32(o.d * SLOT));
        std::string saddr = S2(S_LT);
        uint32_t off_reg = rm0 + 2;
        if (pair && o.negc) {     // the upper half of the wave brings leaf c > a (any leaf): its lanes' offsets are (c - a) leaf strides further
          if (o.a != 0) {
            E.ins("s_mul_i32 " + S(S_X) + ", " + S(S_LS8) + ", " + hex32(o.a));
            E.ins("s_add_u32 " + S(S_FA) + ", " + S(S_LT) + ", " + S(S_X));
            E.ins("s_addc_u32 " + S(S_FA + 1) + ", " + S(S_LT + 1) + ", 0");
            saddr = S2(S_FA);
          }
          E.ins("s_mul_i32 " + S(S_X) + ", " + S(S_LS8) + ", " + hex32(o.c - o.a));
          E.ins("s_sub_u32 " + S(S_X) + ", " + S(S_X) + ", 0x200");
          E.ins("v_add_u32_e32 v" + std::to_string(rm0 + 3) + ", " + S(S_X) + ", v" + std::to_string(rm0 + 2));
          E.ins("v_cndmask_b32_e64 v" + std::to_string(rm0 + 3) + ", v" + std::to_string(rm0 + 2) + ", v" + std::to_string(rm0 + 3) + ", " + S2(S_DELTA + 4));
          off_reg = rm0 + 3;
        } else if (o.a != 0 && !(fdg::knob("FDG_POOL_VADDR") && fdg::knob("FDG_POOL_VADDR")[0] == '1')) {
          // (scalar address arithmetic: measured 1.5 % faster than one vector add into the offset register -- FDG_POOL_VADDR=1 -- although that is
          //  two instructions fewer: the scalar unit's instructions cost a lone wave less than a vector one, profiles/r06_log_sweep_c.txt)
          if (cs->pool_unit == 2) E.ins("s_add_u32 " + S(S_FA) + ", " + S(S_LT) + ", " + hex32(o.a * 512u));
          else {
            E.ins("s_mul_i32 " + S(S_X) + ", " + S(S_LS8) + ", " + hex32(o.a));
            E.ins("s_add_u32 " + S(S_FA) + ", " + S(S_LT) + ", " + S(S_X));
          }
          E.ins("s_addc_u32 " + S(S_FA + 1) + ", " + S(S_LT + 1) + ", 0");
          saddr = S2(S_FA);
        } else if (o.a != 0) {
          if (cs->pool_unit == 2) E.ins("v_add_u32_e32 v" + std::to_string(rm0 + 3) + ", " + hex32(o.a * 512u) + ", v" + std::to_string(rm0 + 2));
          else {
            E.ins("s_mul_i32 " + S(S_X) + ", " + S(S_LS8) + ", " + hex32(o.a));
            E.ins("v_add_u32_e32 v" + std::to_string(rm0 + 3) + ", " + S(S_X) + ", v" + std::to_string(rm0 + 2));
          }
          off_reg = rm0 + 3;
        } else {
          E.ins("s_nop 0");       // (leaf 0: nothing to compute between the write of M0 and the load)
        }
        E.ins("global_load_lds_dwordx4 v" + std::to_string(off_reg) + ", " + saddr + pool_policy);
        pool_pending.push_back({++E.vm_issued, (uint32_t)o.imm});
        const bool more = this_op + 1 < prog.ops.size() && prog.ops[this_op + 1].kind == M_POOL_FETCH;
        if (!more && pool_exec_low) { E.ins("s_mov_b64 exec, -1"); pool_exec_low = false; }
        break;
      }
      case M_SEND:
        E.wait_reg(o.a);
        E.ins(DSW + (o.d < 128 ? V(V_LANE8) : "v" + std::to_string(rm0 + 1)) + ", " + vall(o.a) + " offset:" + std::to_string((o.d % 128) * SLOT));
        ++E.lg_issued;
        break;
      case M_BARRIER: {      // this wave's LDS traffic of the epoch has completed; then all four waves meet
        // pooled programs: what this wave fetched for the epoch that starts behind this barrier has landed in the pool
        uint64_t need_seq = 0;
        for (auto it = pool_pending.begin(); it != pool_pending.end();) {
          if (it->second <= bar_seen + 1) { need_seq = std::max(need_seq, it->first); it = pool_pending.erase(it); } else ++it;
        }
        if (need_seq && !(dbg && std::strstr(dbg, "nofetchwait"))) E.wait_vm(need_seq);     // (timing experiment: results are garbage without it)
        bar_seen++;
        if (cs && cs->slack) {
          // Flag synchronisation (fdg_opt.h: CoopProgram::slack).  Publish first -- the store is queued behind this wave's LDS reads of the epoch,
          // which therefore have read their slots before anyone sees the new count --, then wait until every other wave has reached sync point
          // o.a of this tile (o.a = 0: the tile's last sync point, a real barrier).  The others' words were read at the previous sync point; only
          // when that stale copy is not enough does the wave poll (s_sleep between reads; after 2^14 rounds it goes on, so that a bug shows
          // as a wrong result and not as a hung device).
          const std::string vf = "v" + std::to_string(rm0 + 6), vfa = "v" + std::to_string(rm0 + 5), vme = "v" + std::to_string(rm0 + 4);
          E.ins("s_add_u32 " + S(S_DELTA + 6) + ", " + S(S_DELTA + 6) + ", 1");
          E.ins("v_mov_b32_e32 " + V(V_TMP) + ", " + S(S_DELTA + 6));
          E.ins("ds_write_b32 " + vme + ", " + V(V_TMP));
          ++E.lg_issued;
          if (o.a == 0) {
            E.ins("s_waitcnt lgkmcnt(0)");
            E.lg_done = E.lg_issued;
            for (auto &pp : E.pend) if (pp.first == 2) pp.first = 0;
            E.ins("s_barrier");
          } else {
            const std::string tag = sfx + "_" + std::to_string(bar_seen);
            E.ins("s_add_u32 " + S(S_DELTA + 8) + ", " + S(S_DELTA + 7) + ", " + std::to_string(o.a));          // what the others must have reached
            E.wait_lg_seq(flag_seq);
            E.ins("v_cmp_gt_u32_e32 vcc, " + S(S_DELTA + 8) + ", " + vf);
            E.ins("s_cbranch_vccz .Lgo" + tag);
            E.ins("s_mov_b32 " + S(S_DELTA + 9) + ", 0");
            os << ".Lpoll" << tag << ":\n";
            E.ins("s_sleep 1");
            E.ins("ds_read_b32 " + vf + ", " + vfa);
            E.ins("s_waitcnt lgkmcnt(0)");
            E.ins("v_cmp_gt_u32_e32 vcc, " + S(S_DELTA + 8) + ", " + vf);
            E.ins("s_add_u32 " + S(S_DELTA + 9) + ", " + S(S_DELTA + 9) + ", 1");
            E.ins("s_bitcmp1_b32 " + S(S_DELTA + 9) + ", 14");
            E.ins("s_cbranch_scc1 .Lgo" + tag);
            E.ins("s_cbranch_vccnz .Lpoll" + tag);
            os << ".Lgo" << tag << ":\n";
          }
          E.ins("ds_read_b32 " + vf + ", " + vfa);                      // the others' progress, for the next sync point
          flag_seq = ++E.lg_issued;
          break;
        }
        E.ins("s_waitcnt lgkmcnt(0)");
        E.lg_done = E.lg_issued;
        for (auto &pp : E.pend) if (pp.first == 2) pp.first = 0;
        if (!(dbg && std::strstr(dbg, "nobarrier"))) E.ins("s_barrier");     // (timing experiment: results are garbage without it)
        break;
      }
      case M_ST_LDS:
        if (dbg_nolds) break;
        E.wait_reg(o.a);
        E.ins(DSW + (cs ? "v" + std::to_string(rm0) : V(V_LANE8)) + ", " + vall(o.a) + " offset:" + std::to_string(o.d * SLOT));
        ++E.lg_issued;
        break;
      case M_MUL:
      case M_ADD:
        E.wait_reg(o.a);
        E.wait_reg(o.b);
        E.wait_reg(o.d);
        if (dbg_novalu) break;
        valu2(o.kind == M_MUL ? "v_mul_f64 " : "v_add_f64 ", o, (o.negb ? "-" : "") + vlo(o.b), (o.negb ? "-" : "") + vhi(o.b));
        break;
      case M_MULC: {
        E.wait_reg(o.a);
        E.wait_reg(o.d);
        if (dbg_novalu) break;
        // A factor +-2^k (spin and symmetry factors mostly are: -2, 4, 0.5, ... -- 13 % of the executed ops of the GV 5-loop
        // self-energy): x * 2^k is v_ldexp_f64(x, k) bit for bit -- both are the exactly scaled value rounded once, denormal
        // results and overflow included -- and the sign rides on the source modifier.  The exponent adder instead of the
        // multiplier array: the graphs that run against the power budget clock higher.  FDG_ISA_NO_LDEXP=1: v_mul_f64.
        int k2 = 0;
        const double m = std::frexp(std::fabs(o.imm), &k2);      // |imm| = m * 2^k2, m in [0.5, 1)
        if (use_ldexp && !o.param && std::isfinite(o.imm) && o.imm != 0.0 && m == 0.5 && k2 - 1 >= -16 && k2 - 1 <= 64) {
          const bool neg = (o.nega != 0) != std::signbit(o.imm);
          E.ins("v_ldexp_f64 " + vlo(o.d) + ", " + (neg ? "-" : "") + vlo(o.a) + ", " + std::to_string(k2 - 1));
          if (W == 2) E.ins("v_ldexp_f64 " + vhi(o.d) + ", " + (neg ? "-" : "") + vhi(o.a) + ", " + std::to_string(k2 - 1));
          break;
        }
        const std::string c = op_const(o);
        valu2("v_mul_f64 ", o, c, c);
        break;
      }
      case M_ADDC: {
        E.wait_reg(o.a);
        E.wait_reg(o.d);
        const std::string c = op_const(o);
        valu2("v_add_f64 ", o, c, c);
        break;
      }
      case M_EXP: {
        E.wait_reg(o.a);
        E.wait_reg(o.d);
        const std::string x = (o.nega ? "-" : "") + vlo(o.a);
        E.ins("v_mul_f64 " + tA + ", " + x + ", " + const_operand(kLog2e));
        E.ins("v_rndne_f64_e32 " + tA + ", " + tA);
        E.ins("v_fma_f64 " + tB + ", " + tA + ", " + const_operand(-kLn2Hi) + ", " + x);
        E.ins("v_fma_f64 " + tB + ", " + tA + ", " + const_operand(-kLn2Lo) + ", " + tB);
        // (the argument register is dead from here: d may be it)
        E.ins("v_mul_f64 " + vlo(o.d) + ", " + tB + ", " + const_operand(kExpC[kExpDeg]));
        E.ins("v_add_f64 " + vlo(o.d) + ", " + vlo(o.d) + ", " + const_operand(kExpC[kExpDeg - 1]));
        for (int k = kExpDeg - 2; k >= 0; --k) E.ins("v_fma_f64 " + vlo(o.d) + ", " + vlo(o.d) + ", " + tB + ", " + const_operand(kExpC[k]));
        E.ins("v_cvt_i32_f64_e32 " + tAd(0) + ", " + tA);
        E.ins("v_ldexp_f64 " + vlo(o.d) + ", " + vlo(o.d) + ", " + tAd(0));
        break;
      }
      case M_RCP: {
        E.wait_reg(o.a);
        E.wait_reg(o.d);
        const std::string x = (o.nega ? "-" : "") + vlo(o.a), mx = (o.nega ? "" : "-") + vlo(o.a);
        E.ins("v_rcp_f64_e64 " + tA + ", " + x);
        E.ins("v_fma_f64 " + tB + ", " + mx + ", " + tA + ", 1.0");
        E.ins("v_fma_f64 " + tA + ", " + tA + ", " + tB + ", " + tA);
        E.ins("v_fma_f64 " + tB + ", " + mx + ", " + tA + ", 1.0");
        E.ins("v_fma_f64 " + vlo(o.d) + ", " + tA + ", " + tB + ", " + tA);
        break;
      }
      case M_SEL: {   // d = cond(c) ? a : b
        E.wait_reg(o.a);
        E.wait_reg(o.b);
        E.wait_reg(o.c);
        E.wait_reg(o.d);
        if (o.imm == 2.0) {      // isfinite: classes -normal .. +normal (bits 3..8)
          E.ins("s_mov_b32 " + S(S_C) + ", 0x1f8");
          E.ins("v_cmp_class_f64_e64 vcc, " + vlo(o.c) + ", " + S(S_C));
        } else
        E.ins(std::string(o.imm != 0.0 ? "v_cmp_ge_f64_e64" : "v_cmp_gt_f64_e64") + " vcc, " + (o.negc ? "-" : "") + vlo(o.c) + ", 0");
        std::string ahi = vd(o.a, 1), bhi = vd(o.b, 1);
        if (o.nega) { E.ins("v_xor_b32_e32 " + tAd(0) + ", 0x80000000, " + ahi); ahi = tAd(0); }
        if (o.negb) { E.ins("v_xor_b32_e32 " + tAd(1) + ", 0x80000000, " + bhi); bhi = tAd(1); }
        E.ins("v_cndmask_b32_e32 " + vd(o.d, 0) + ", " + vd(o.b, 0) + ", " + vd(o.a, 0) + ", vcc");
        E.ins("v_cndmask_b32_e32 " + vd(o.d, 1) + ", " + bhi + ", " + ahi + ", vcc");
        break;
      }
      case M_FMAK: {  // d = a * b + imm, one rounding (pow_body's fma with a constant addend)
        E.wait_reg(o.a);
        E.wait_reg(o.b);
        E.wait_reg(o.d);
        E.ins("v_fma_f64 " + vlo(o.d) + ", " + (o.nega ? "-" : "") + vlo(o.a) + ", " + (o.negb ? "-" : "") + vlo(o.b) + ", " + op_const(o));
        break;
      }
      case M_CONST: {
        E.wait_reg(o.d);
        uint64_t u;
        std::memcpy(&u, &o.imm, 8);
        E.ins("v_mov_b32_e32 " + vd(o.d, 0) + ", " + hex32((uint32_t)u));
        E.ins("v_mov_b32_e32 " + vd(o.d, 1) + ", " + hex32((uint32_t)(u >> 32)));
        break;
      }
      case M_DIV1: {  // d = 1.0 / a, correctly rounded: the sequence the compiler emits for an fp64 division
        E.wait_reg(o.a);
        E.wait_reg(o.d);
        const std::string x = vlo(o.a);
        E.ins("v_div_scale_f64 " + tA + ", " + S2(S_X) + ", " + x + ", " + x + ", 1.0");
        E.ins("v_rcp_f64_e32 " + tB + ", " + tA);
        E.ins("v_div_scale_f64 " + tC + ", vcc, 1.0, " + x + ", 1.0");
        E.ins("v_fma_f64 " + tD + ", -" + tA + ", " + tB + ", 1.0");
        E.ins("v_fma_f64 " + tB + ", " + tB + ", " + tD + ", " + tB);
        E.ins("v_fma_f64 " + tD + ", -" + tA + ", " + tB + ", 1.0");
        E.ins("v_fma_f64 " + tB + ", " + tB + ", " + tD + ", " + tB);
        E.ins("v_mul_f64 " + tD + ", " + tC + ", " + tB);
        E.ins("v_fma_f64 " + tA + ", -" + tA + ", " + tD + ", " + tC);
        E.ins("v_div_fmas_f64 " + tA + ", " + tA + ", " + tB + ", " + tD);
        E.ins("v_div_fixup_f64 " + vlo(o.d) + ", " + tA + ", " + x + ", 1.0");
        break;
      }
      case M_FIXZ: {  // d = a == 0 ? imm : a
        E.wait_reg(o.a);
        E.wait_reg(o.d);
        const int c = op_const_sgpr(o);
        E.ins("v_cmp_eq_f64_e64 vcc, " + vlo(o.a) + ", 0");
        E.ins("v_mov_b32_e32 " + tAd(0) + ", " + S(c));
        E.ins("v_mov_b32_e32 " + tAd(1) + ", " + S(c + 1));
        E.ins("v_cndmask_b32_e32 " + vd(o.d, 0) + ", " + vd(o.a, 0) + ", " + tAd(0) + ", vcc");
        E.ins("v_cndmask_b32_e32 " + vd(o.d, 1) + ", " + vd(o.a, 1) + ", " + tAd(1) + ", vcc");
        break;
      }
      case M_SELC: {  // d = cond(a) ? imm : -imm
        E.wait_reg(o.a);
        E.wait_reg(o.d);
        const int c = op_const_sgpr(o);
        E.ins(std::string(o.negb ? "v_cmp_ge_f64_e64" : "v_cmp_gt_f64_e64") + " vcc, " + (o.nega ? "-" : "") + vlo(o.a) + ", 0");
        E.ins("v_mov_b32_e32 " + vd(o.d, 0) + ", " + S(c));
        E.ins("v_mov_b32_e32 " + tAd(0) + ", " + S(c + 1));
        E.ins("v_xor_b32_e32 " + tAd(1) + ", 0x80000000, " + tAd(0));
        E.ins("v_cndmask_b32_e32 " + vd(o.d, 1) + ", " + tAd(1) + ", " + tAd(0) + ", vcc");
        break;
      }
      case M_FMA:     // FDG_SPEC_FAST_MATH only
      case M_FMAC: {
        E.wait_reg(o.a);
        if (o.kind == M_FMA) E.wait_reg(o.b);
        E.wait_reg(o.c);
        E.wait_reg(o.d);
        const std::string k = o.kind == M_FMAC ? op_const(o) : std::string();
        for (int h = 0; h < W; ++h) {
          auto part = [&](uint32_t r) { return h ? vhi(r) : vlo(r); };
          E.ins("v_fma_f64 " + part(o.d) + ", " + (o.nega ? "-" : "") + part(o.a) + ", " +
                (o.kind == M_FMA ? (o.negb ? "-" : "") + part(o.b) : k) + ", " + (o.negc ? "-" : "") + part(o.c));
        }
        break;
      }
      case M_LD_ACC:
        if (dbg_noacc) break;
        E.wait_reg(o.d);
        if (o.a < E.pend_acc.size() && E.pend_acc[o.a]) { E.wait_vm(E.pend_acc[o.a]); E.pend_acc[o.a] = 0; }     // a landing slot: the load must have arrived
        for (int k = 0; k < RW; ++k)
          E.ins("v_accvgpr_read_b32 v" + std::to_string(V_BASE + RW * o.d + k) + ", a" + std::to_string(RW * o.a + k));
        break;
      case M_ST_ACC:
        if (dbg_noacc) break;
        E.wait_reg(o.a);
        for (int k = 0; k < RW; ++k)
          E.ins("v_accvgpr_write_b32 a" + std::to_string(RW * o.d + k) + ", v" + std::to_string(V_BASE + RW * o.a + k));
        break;
      case M_ROOT: {
        E.wait_reg(o.a);
        if (accumulate) {   // acc_k = acc_k + w * root_k
          if (w_seq > E.vm_done) {
            const uint64_t n = std::min<uint64_t>(E.vm_issued - w_seq, 63);
            E.ins("s_waitcnt vmcnt(" + std::to_string(n) + ")");
            E.vm_done = std::max(E.vm_done, std::max(E.vm_issued - n, w_seq));
          }
          E.ins("v_mul_f64 " + vtmp + ", " + vwgt + ", " + (o.nega ? "-" : "") + vlo(o.a));
          if (aa) {
            acc_to_tmp2(o.d);
            E.ins("v_add_f64 " + vtmp2 + ", " + vtmp2 + ", " + vtmp);
            tmp2_to_acc(o.d);
            break;
          }
          E.ins("v_add_f64 " + vacc(o.d) + ", " + vacc(o.d) + ", " + vtmp);
          break;
        }
        emit_scaled_addr(E, S_A, S_RT, S_RK8, o.d);
        for (int h = 0; h < W; ++h) {
          const int src = V_BASE + RW * o.a + 2 * h;
          if (h == 1) {   // second sample of the lane: one row further
            E.ins("s_lshl_b64 " + S2(S_T) + ", " + S2(S_RS) + ", 3");
            E.ins("s_add_u32 " + S(S_A) + ", " + S(S_A) + ", " + S(S_T));
            E.ins("s_addc_u32 " + S(S_A + 1) + ", " + S(S_A + 1) + ", " + S(S_T + 1));
          }
          if (o.nega) {
            E.ins("v_mov_b32_e32 " + V(V_TMP) + ", v" + std::to_string(src));
            E.ins("v_xor_b32_e32 " + V(V_TMP + 1) + ", 0x80000000, v" + std::to_string(src + 1));
            E.ins("global_store_dwordx2 " + V(V_ROOTOFF) + ", v[" + std::to_string(V_TMP) + ":" + std::to_string(V_TMP + 1) + "], " + S2(S_A) + root_policy);
          } else {
            E.ins("global_store_dwordx2 " + V(V_ROOTOFF) + ", v[" + std::to_string(src) + ":" + std::to_string(src + 1) + "], " + S2(S_A) + root_policy);
          }
          ++E.vm_issued;
        }
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
  if (tile_run) {
    E.ins("s_add_u32 " + S(S_TILE) + ", " + S(S_TILE) + ", 1");
    E.ins("s_and_b32 " + S(S_X) + ", " + S(S_TILE) + ", " + std::to_string((1 << tile_run) - 1));
    E.ins("s_cbranch_scc1 .Lrun" + sfx);
    E.ins("s_lshl_b32 " + S(S_X) + ", " + S(S_NWG) + ", " + std::to_string(tile_run));
    E.ins("s_add_u32 " + S(S_TILE) + ", " + S(S_TILE) + ", " + S(S_X));
    E.ins("s_sub_u32 " + S(S_TILE) + ", " + S(S_TILE) + ", " + std::to_string(1 << tile_run));
    os << ".Lrun" << sfx << ":\n";
  } else if (tile_xcd) {
    E.ins("s_lshr_b32 " + S(S_X) + ", " + S(S_NWG) + ", 3");
    E.ins("s_add_u32 " + S(S_TILE) + ", " + S(S_TILE) + ", " + S(S_X));
  } else if (tile_rot) {
    E.ins("s_add_u32 " + S(S_ROTB) + ", " + S(S_ROTB) + ", " + S(S_NWG));
    E.ins("s_add_u32 " + S(S_ROTS) + ", " + S(S_ROTS) + ", " + std::to_string(tile_rot));
    E.ins("s_sub_u32 " + S(S_X) + ", " + S(S_NWG) + ", 1");
    E.ins("s_and_b32 " + S(S_ROTS) + ", " + S(S_ROTS) + ", " + S(S_X));
    E.ins("s_add_u32 " + S(S_TILE) + ", " + S(S_ROTB) + ", " + S(S_ROTS));
  } else {
    E.ins("s_add_u32 " + S(S_TILE) + ", " + S(S_TILE) + ", " + S(S_NWG));
  }
  E.ins("s_cmp_ge_u32 " + S(S_TILE) + ", " + S(S_NTILES));
  E.ins("s_cbranch_scc0 .Lback" + sfx);
  if (accumulate) {
    // butterfly sum over the 64 lanes (IEEE addition commutes, so every lane ends with the same bits),
    // then partial[k][wg] = lane 0's value; fdg_reduce_lane_partials adds the waves in fixed order
    E.ins("s_mov_b64 exec, -1");
    E.ins("v_lshlrev_b32_e32 " + V(V_TMP) + ", 2, v0");
    if (aa)
      for (uint32_t k = 0; k < p.R; ++k) {          // root by root: accumulator -> VGPR pair, butterfly, back
        acc_to_tmp2(k);
        for (int off = 32; off >= 1; off >>= 1) {
          E.ins("v_xor_b32_e32 " + V(V_TMP + 1) + ", " + std::to_string(off * 4) + ", " + V(V_TMP));
          E.ins("ds_bpermute_b32 v" + std::to_string(wgt0 + 2) + ", " + V(V_TMP + 1) + ", v" + std::to_string(wgt0 + 4));
          E.ins("ds_bpermute_b32 v" + std::to_string(wgt0 + 3) + ", " + V(V_TMP + 1) + ", v" + std::to_string(wgt0 + 5));
          E.ins("s_waitcnt lgkmcnt(0)");
          E.ins("v_add_f64 " + vtmp2 + ", " + vtmp2 + ", " + vtmp);
        }
        tmp2_to_acc(k);
      }
    for (int off = 32; off >= 1 && !aa; off >>= 1) {
      E.ins("v_xor_b32_e32 " + V(V_TMP + 1) + ", " + std::to_string(off * 4) + ", " + V(V_TMP));
      for (uint32_t k = 0; k < p.R; ++k) {
        const uint32_t a = acc0 + 2 * k, t = acc0 + 2 * (p.R + 1);
        E.ins("ds_bpermute_b32 v" + std::to_string(t) + ", " + V(V_TMP + 1) + ", v" + std::to_string(a));
        E.ins("ds_bpermute_b32 v" + std::to_string(t + 1) + ", " + V(V_TMP + 1) + ", v" + std::to_string(a + 1));
        E.ins("s_waitcnt lgkmcnt(0)");
        E.ins("v_add_f64 " + vacc(k) + ", " + vacc(k) + ", " + vtmp);
      }
    }
    E.ins("s_mov_b64 exec, 1");
    E.ins("v_mov_b32_e32 " + V(V_TMP) + ", 0");
    E.ins("s_lshl_b32 " + S(S_X) + ", s2, 3");
    E.ins("s_add_u32 " + S(S_A) + ", " + S(S_ROOT) + ", " + S(S_X));
    E.ins("s_addc_u32 " + S(S_A + 1) + ", " + S(S_ROOT + 1) + ", 0");
    E.ins("s_lshl_b32 " + S(S_T) + ", " + S(S_NWG) + ", 3");
    for (uint32_t k = 0; k < p.R; ++k) {
      if (aa) acc_to_tmp2(k);
      E.ins("global_store_dwordx2 " + V(V_TMP) + ", " + (aa ? vtmp2 : vacc(k)) + ", " + S2(S_A));
      E.ins("s_add_u32 " + S(S_A) + ", " + S(S_A) + ", " + S(S_T));
      E.ins("s_addc_u32 " + S(S_A + 1) + ", " + S(S_A + 1) + ", 0");
    }
  }
  E.ins("s_endpgm");
  // long backward jump (the body may exceed the 16-bit branch range)
  os << ".Lback" << sfx << ":\n";
  E.ins("s_getpc_b64 " + S2(S_A));
  os << ".Lpc" << sfx << ":\n";
  E.ins("s_add_u32 " + S(S_A) + ", " + S(S_A) + ", (.Ltile" + sfx + "-.Lpc" + sfx + ")&0xffffffff");
  E.ins("s_addc_u32 " + S(S_A + 1) + ", " + S(S_A + 1) + ", (.Ltile" + sfx + "-.Lpc" + sfx + ")>>32");
  E.ins("s_setpc_b64 " + S2(S_A));
  E.ins("s_endpgm");

  // ---- kernel descriptor -------------------------------------------------------
  const uint32_t next_vgpr = std::max<uint32_t>(V_BASE + RW * std::max<uint32_t>(prog.n_reg_used, 1) + 2 * acc_pairs_v + 2 * n_tmp_pairs + (rm_bufs ? 10 : 0) + (cs ? (cs->slack ? 7 : 4) : 0) + (rl ? 2 : 0), 8);
  const uint32_t accum = (next_vgpr + 3) & ~3u;
  const uint32_t n_agpr = RW * prog.n_acc_used + (aa ? 2 * p.R : 0);
  if (cs) return KernelMeta{kname, lds_bytes, accum, n_agpr, 12, (mc || has_macro) ? S_END : S_POOL + 2 * 16};
  os << "\t.section\t.rodata,\"a\",@progbits\n\t.p2align\t6, 0x0\n\t.amdhsa_kernel " << kname << "\n";
  os << "\t\t.amdhsa_group_segment_fixed_size " << lds_bytes << "\n";
  os << "\t\t.amdhsa_private_segment_fixed_size 0\n\t\t.amdhsa_kernarg_size " << (mc ? 144 : 96) << "\n\t\t.amdhsa_user_sgpr_count 2\n";
  os << "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1\n\t\t.amdhsa_system_sgpr_workgroup_id_x 1\n";
  os << "\t\t.amdhsa_system_vgpr_workitem_id 0\n";
  const int n_sgpr = (mc || has_macro) ? S_END : S_POOL + 2 * 16;   // programs without leaf formulas: the 16-entry pool ends the map
  os << "\t\t.amdhsa_next_free_vgpr " << (accum + n_agpr) << "\n\t\t.amdhsa_next_free_sgpr " << n_sgpr << "\n";
  os << "\t\t.amdhsa_accum_offset " << accum << "\n\t\t.amdhsa_reserve_vcc 1\n";
  os << "\t\t.amdhsa_float_round_mode_32 0\n\t\t.amdhsa_float_round_mode_16_64 0\n";
  os << "\t\t.amdhsa_float_denorm_mode_32 3\n\t\t.amdhsa_float_denorm_mode_16_64 3\n";
  os << "\t\t.amdhsa_dx10_clamp 1\n\t\t.amdhsa_ieee_mode 1\n";
  os << "\t.end_amdhsa_kernel\n";
  (void)p;
  return KernelMeta{kname, lds_bytes, accum, n_agpr, mc ? 18 : 12, n_sgpr};
}

}  // namespace

// The cooperative kernel: a workgroup of four waves (one per SIMD: the register and LDS requests leave room for nothing else
// on the CU); the entry reads the wave's number and jumps to that wave's section.
static KernelMeta emit_coop(Emit &E, const Lowered &p, const CoopProgram &cp, const std::string &kname) {
  std::ostringstream &os = E.os;
  os << "\t.text\n\t.protected\t" << kname << "\n\t.globl\t" << kname << "\n\t.p2align\t8\n\t.type\t" << kname << ",@function\n";
  os << kname << ":\n";
  E.off = 0;
  E.klabel = kname;
  // Alignment pads (round 6): in the POOLED kernels, in front of vector instructions only (Emit::align 2) -- gv_ver4_4 3.12 -> 2.95 ms, +5.6 %
  // (every 8-byte instruction: 3.01; profiles/r06_log_sweep_f.txt).  Round 5 measured -12 % and left them out: the printer's offsets were four
  // bytes off behind the prologue's `s_mov_b32 s, 0xffffffff` (the assembler's inline constant -1, counted as a literal), so every pad sat in the
  // wrong place (tests/test_isa_alignment.py now lets the assembler check the offsets).  The value-passing cooperative kernels lose 2 % with
  // the pads and stay without.  FDG_COOP_ALIGN=0 / 1 forces.
  const char *ca = fdg::knob("FDG_COOP_ALIGN");
  E.pad_ok = ca ? ca[0] == '1' : cp.pooled;
  const int align_before = E.align;
  if (cp.pooled && !ca && E.align == 1) E.align = 2;
  E.hz.reset();
  E.ins("v_lshrrev_b32_e32 v1, 6, v0");
  E.ins("v_readfirstlane_b32 s3, v1");
  for (uint32_t w = 0; w < cp.n_wave; ++w) {
    const std::string lab = ".Lsec_" + kname + "_w" + std::to_string(w), here = ".Lpc_" + kname + "_d" + std::to_string(w);
    E.ins("s_cmp_lg_u32 s3, " + std::to_string(w));
    E.ins("s_cbranch_scc1 .Lnot_" + kname + "_" + std::to_string(w));
    E.ins("s_getpc_b64 s[4:5]");
    os << here << ":\n";
    E.ins("s_add_u32 s4, s4, (" + lab + "-" + here + ")&0xffffffff");
    E.ins("s_addc_u32 s5, s5, (" + lab + "-" + here + ")>>32");
    E.ins("s_setpc_b64 s[4:5]");
    os << ".Lnot_" << kname << "_" << w << ":\n";
  }
  E.ins("s_endpgm");
  uint32_t panel_wg = 0, prefix[CoopProgram::MAXW];
  for (uint32_t w = 0; w < cp.n_wave; ++w) { prefix[w] = panel_wg; panel_wg += std::max<uint32_t>(cp.wave[w].n_mem_used, 1) * 512u; }
  uint32_t accum = 0, n_agpr = 0;
  int n_sgpr = 0;
  for (uint32_t w = 0; w < cp.n_wave; ++w) {
    const CoopSec cs{w, cp.n_shared, (cp.n_shared + w * cp.n_priv_lds) * 512u, panel_wg, prefix[w], cp.pooled, cp.pool_unit,
                     cp.pooled ? cp.slack : 0u, (cp.n_shared + cp.n_wave * cp.n_priv_lds) * 512u, cp.n_wave};
    E.hz.reset();
    const KernelMeta m = emit_kernel(E, p, cp.wave[w], kname + "_w" + std::to_string(w), 1, false, 0, &cs);
    accum = std::max(accum, m.accum); n_agpr = std::max(n_agpr, m.n_agpr); n_sgpr = std::max(n_sgpr, m.n_sgpr);
  }
  const uint32_t lds_bytes = (cp.n_shared + cp.n_wave * cp.n_priv_lds) * 512u + (cp.pooled && cp.slack ? 256u * cp.n_wave : 0u);
  E.align = align_before;
  os << "\t.section\t.rodata,\"a\",@progbits\n\t.p2align\t6, 0x0\n\t.amdhsa_kernel " << kname << "\n";
  os << "\t\t.amdhsa_group_segment_fixed_size " << lds_bytes << "\n";
  os << "\t\t.amdhsa_private_segment_fixed_size 0\n\t\t.amdhsa_kernarg_size 96\n\t\t.amdhsa_user_sgpr_count 2\n";
  os << "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1\n\t\t.amdhsa_system_sgpr_workgroup_id_x 1\n";
  os << "\t\t.amdhsa_system_vgpr_workitem_id 0\n";
  os << "\t\t.amdhsa_next_free_vgpr " << (accum + n_agpr) << "\n\t\t.amdhsa_next_free_sgpr " << n_sgpr << "\n";
  os << "\t\t.amdhsa_accum_offset " << accum << "\n\t\t.amdhsa_reserve_vcc 1\n";
  os << "\t\t.amdhsa_float_round_mode_32 0\n\t\t.amdhsa_float_round_mode_16_64 0\n";
  os << "\t\t.amdhsa_float_denorm_mode_32 3\n\t\t.amdhsa_float_denorm_mode_16_64 3\n";
  os << "\t\t.amdhsa_dx10_clamp 1\n\t\t.amdhsa_ieee_mode 1\n";
  os << "\t.end_amdhsa_kernel\n";
  return KernelMeta{kname, lds_bytes, accum, n_agpr, 12, n_sgpr};
}

// What the row-major variant of `prog` would move with `bufs` staging buffers: chunk fetches (8 KB each) and gathered leaves.
void rm_plan_stats(const Lowered &p, const OptProgram &prog, uint32_t bufs, uint64_t &fetches, uint64_t &gathers) {
  std::vector<std::vector<RmFetch>> f;
  std::vector<int> ld;
  rm_plan(p, prog, bufs, f, ld);
  fetches = gathers = 0;
  for (const auto &v : f) fetches += v.size();
  for (size_t q = 0; q < prog.ops.size(); ++q) if (prog.ops[q].kind == M_LD_LEAF && ld[q] < 0) gathers++;
}

// Re-parses an assembly listing against the hazard table (independent of how the listing was produced).  Returns the
// number of violations; `report` gets one line per violation and a summary of what was checked.
int check_isa_hazards(const std::string &text, std::string &report) {
  HzTracker hz;
  std::istringstream in(text);
  std::string line;
  std::ostringstream rep;
  long lineno = 0, n_inst = 0, n_bad = 0, n_nop = 0, n_trans = 0, n_sgpr_w = 0;
  while (std::getline(in, line)) {
    ++lineno;
    if (!hz_is_inst(line)) continue;
    const HzInst I = hz_decode(line);
    ++n_inst;
    if (I.nop) n_nop++;
    if (I.trans) n_trans++;
    if (I.valu) for (int r : I.wr) if (r >= HR_SGPR) { n_sgpr_w++; break; }
    const char *why = nullptr;
    const int need = hz.missing(I, &why);
    if (need > 0) {
      ++n_bad;
      if (n_bad <= 50) rep << "line " << lineno << ": " << line << "   ; " << need << " more wait state(s): " << (why ? why : "") << "\n";
    }
    hz.issue(I);
  }
  rep << "checked " << n_inst << " instructions (" << n_trans << " transcendental, " << n_sgpr_w << " VALU writes of SGPR/VCC, " << n_nop
      << " s_nop) against " << (sizeof(kHazards) / sizeof(kHazards[0])) << " rules: " << n_bad << " violation(s)\n";
  report = rep.str();
  return (int)n_bad;
}

std::string isa_hazard_table() {
  std::ostringstream os;
  for (const HzRule &R : kHazards) os << R.wait << " wait state(s): " << R.what << "\n";
  return os.str();
}

// One code object: the W = 1 kernel `kname`, and, when prog2 is given, the two-samples-per-lane kernel
// `kname`_w2 next to it.
std::string emit_isa(const Lowered &p, const OptProgram &prog, const std::string &kname, const OptProgram *prog2,
                     const OptProgram *prog_acc, const OptProgram *prog_rm, uint32_t rm_bufs, const CoopProgram *coop,
                     const OptProgram *prog_rm_acc, const CoopProgram *pool, const OptProgram *prog_rl, const OptProgram *prog_rl_acc) {
  Emit E;
  { const char *a = fdg::knob("FDG_ISA_ALIGN"); E.align = a && a[0] >= '0' && a[0] <= '2' ? a[0] - '0' : 1; }
  { const char *a = fdg::knob("FDG_ISA_SHIFT"); E.shift = a ? std::atoi(a) : 0; }
  E.check_off = fdg::knob("FDG_ISA_CHECK_OFF") != nullptr;
  E.os << "\t.amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"\n\t.amdhsa_code_object_version 6\n";
  std::vector<KernelMeta> ks;
  // Programs too long for a second copy (below) get their ONE kernel with the streaming policy (round 6): a leaf's last load of the tile and the
  // root stores non-temporal.  On a batch whose tiles are not whole cache lines that costs a second fetch of the lines two tiles share; the plain
  // policy costs these graphs -- thousands of leaves, re-loaded several times over -- far more on every batch (gv_ver4_4, one-wave: 4.24 -> 3.49 ms
  // with every leaf load non-temporal, profiles/r06_log_pool_sweep2.txt; gv_sigma6 4.61 -> 4.35, profiles/r06_log_nt_sweep.txt).
  const bool long_program = prog.ops.size() > 60000 && prog.mc_n_k == 0 && prog.mc_n_t == 0 && !fdg::knob("FDG_ISA_NO_STREAMING");
  E.streaming = long_program;
  E.nt_dist = long_program ? 512 : -1;
  ks.push_back(emit_kernel(E, p, prog, kname, 1));
  E.streaming = false;
  if (prog2) ks.push_back(emit_kernel(E, p, *prog2, kname + "_w2", 2));
  if (prog_acc) {
    E.streaming = long_program && prog_acc->ops.size() > 60000;
    ks.push_back(emit_kernel(E, p, *prog_acc, kname + "_acc", 1, true));
    E.streaming = false;
  }
  E.nt_dist = -1;
  // the same programs once more for batches whose tiles are whole cache lines (see `streaming` in emit_kernel); not for programs
  // so long that a second copy would double a minute of assembly
  if (!fdg::knob("FDG_ISA_NO_STREAMING") && prog.ops.size() <= 60000 && prog.mc_n_k == 0 && prog.mc_n_t == 0) {
    E.streaming = true;
    ks.push_back(emit_kernel(E, p, prog, kname + "_nt", 1));
    if (prog_acc && prog_acc->ops.size() <= 60000) ks.push_back(emit_kernel(E, p, *prog_acc, kname + "_acc_nt", 1, true));
    E.streaming = false;
  }
  if (prog_rm && rm_bufs) ks.push_back(emit_kernel(E, p, *prog_rm, kname + "_rm", 1, false, rm_bufs));
  if (prog_rm_acc && rm_bufs) ks.push_back(emit_kernel(E, p, *prog_rm_acc, kname + "_rm_acc", 1, true, rm_bufs));
  if (prog_rl) ks.push_back(emit_kernel(E, p, *prog_rl, kname + "_rl", 1, false, 0, nullptr, true));
  if (prog_rl && prog_rl_acc) ks.push_back(emit_kernel(E, p, *prog_rl_acc, kname + "_rl_acc", 1, true, 0, nullptr, true));
  if (coop && coop->supported) { ks.push_back(emit_coop(E, p, *coop, kname + "_coop")); ks.back().wg = 64 * coop->n_wave; }
  if (pool && pool->supported) { ks.push_back(emit_coop(E, p, *pool, kname + "_pool")); ks.back().wg = 64 * pool->n_wave; }
  std::ostringstream &os = E.os;
  os << "\t.text\n\t.amdgpu_metadata\n---\namdhsa.kernels:\n";
  const char *kinds[18] = {"global_buffer", "by_value", "by_value", "global_buffer", "by_value", "by_value",
                           "global_buffer", "by_value", "by_value", "global_buffer", "global_buffer", "by_value",
                           "by_value", "by_value", "by_value", "by_value", "by_value", "by_value"};
  for (const KernelMeta &k : ks) {
    os << "  - .agpr_count: " << k.n_agpr << "\n    .args:\n";
    for (int i = 0; i < k.n_args; ++i) {
      const char *kind = (k.n_args == 12 && i >= 10) ? "by_value" : kinds[i];      // (lts, rts end every argument list)
      os << "      - .offset: " << i * 8 << "\n        .size: 8\n        .value_kind: " << kind << "\n";
      if (std::strcmp(kind, "global_buffer") == 0) os << "        .address_space: global\n";
    }
    os << "    .group_segment_fixed_size: " << k.lds_bytes << "\n    .kernarg_segment_align: 8\n    .kernarg_segment_size: " << 8 * k.n_args << "\n";
    os << "    .max_flat_workgroup_size: " << k.wg << "\n    .name: " << k.name << "\n    .private_segment_fixed_size: 0\n";
    os << "    .sgpr_count: " << (k.n_sgpr + 6) << "\n    .sgpr_spill_count: 0\n    .symbol: " << k.name << ".kd\n";
    os << "    .uniform_work_group_size: 1\n    .uses_dynamic_stack: false\n    .vgpr_count: " << (k.accum + k.n_agpr)
       << "\n    .vgpr_spill_count: 0\n    .wavefront_size: 64\n";
  }
  os << "amdhsa.target: amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\t.end_amdgpu_metadata\n";
  return os.str();
}

}  // namespace fdg
