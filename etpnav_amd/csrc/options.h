// Run-time switches of libetpnav_hip.so: ONE table, read from the environment once (ETP_<NAME>, at the first lookup) and changed
// afterwards only through the C ABI (etp_option_set; include/etpnav_hip.h).  Until round 5 every switch was its own getenv() --
// three of them on every GEMM launch.  A lookup is an array read.
// Switches whose results are WRONG by design (timing-only drop-one builds: SKIP_LN / SKIP_ATTN / SKIP_WGRAD) are not in the shipped
// library any more: they exist only under -DETP_EXPERIMENTS.
#pragma once

namespace etp {

enum Opt {
  OPT_MM32,             // 0: mm32 family off | 128 / 64 / 264: force a class for every eligible product (tests, A/B runs)
  OPT_MM32_GROUP,       // 128 / 256: force the grouped weight-gradient class
  OPT_MM32_K2,          // 0: one-round 128x64 grids keep four wavefronts | 262 / 264: split-reduction form with rings of two / three
  OPT_GEMM_TILE,        // gemm.hip tile class: "128", "64", "32", "w", "256" + optional "s2".."s4", "r" = register-staged
  OPT_GROUP_TILE,       // gemm.hip grouped class: "256s2", "256s3", "128s2", "128s3", "64s3", "64s4"
  OPT_GEMM_WIDE,        // 1: gemm.hip's 128x64 class on
  OPT_GEMM_SMALL,       // 0: gemm.hip's 32x64 class off
  OPT_GEMM_XCD,         // 0: XCD-chunked tile order off
  OPT_ATTN_FUSED, OPT_ATTN_FLASH, OPT_ATTN_Q96, OPT_ATTN_ROWS,
  OPT_LNBWD_GRID, OPT_LNBWD_TWO_STAGE,
  OPT_WGRAD_GROUP, OPT_FLUSH_DELAY, OPT_FLUSH_EVERY,
  OPT_ROW_EXCLUSIVE,    // 0: the row kernels of DESIGN.md §3.6 launch WITHOUT the CU-exclusive LDS request (neighbour-matrix test only)
  OPT_ATTN_PROJ,        // out-projection dgrad folded into the register-resident attention backward: unset = when batch*heads <= CUs, 0 never, 1 always
  OPT_NAV_TAIL,         // bit 0: the node-embedding backward as a leaf on the weight-gradient stream; bit 1: d txt_embeds joined by its consumers (default 3)
  OPT_TXT_LAST_SPLIT,   // text layer 0's attention weight gradients forked each as soon as it can: unset = for multi-round grids (config 4), 0 never, 1 always
  OPT_TXT_TAIL,         // 1: the text-embedding backward on the aux2 stream beside the weight-gradient backlog (measured neutral: default off)
  OPT_ATTN_QKV,         // QKV projection folded into the register-resident self-attention forward: unset / 0 never (measured slower), 1 always
#ifdef ETP_EXPERIMENTS
  OPT_SKIP_LN, OPT_SKIP_ATTN, OPT_SKIP_WGRAD,
#endif
  OPT_COUNT
};

const char* opt_str(Opt o);             // nullptr when unset (or set to the empty string)
int opt_int(Opt o, int dflt);           // atoi of the value, dflt when unset
inline bool opt_on(Opt o, bool dflt) {  // "0" = off, anything else = on
  const char* s = opt_str(o);
  return s ? s[0] != '0' : dflt;
}
int opt_set(const char* name, const char* value);   // name without the ETP_ prefix; value NULL / "" = unset.  0 ok, -1 unknown name
int opt_get(const char* name, char* out, int cap);  // length of the value (0 = unset), -1 unknown name
int opt_list(char* out, int cap);                   // "NAME=value\n" for every switch that is set; returns the length needed

}  // namespace etp
