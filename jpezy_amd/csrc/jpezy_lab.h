// jpezy_lab.h -- the LABORATORY of the f32 encode path: included by jpezy_f32_quad.h only in builds with -DJPEZY_WITH_LAB
// (tools/ab/ab_build.py, `python -m jpezy_amd._build --lab`).  The shipped library is built without it and contains neither the
// persistent kernels (encode variants 2 and 3, jpezy_kernels_f32_ps.hip: built, parity-green and measured NOT faster in round 5,
// docs/ROUND5.md) nor any of the switches below.
//   PROBE_*         timing probes that leave the results unchanged: N extra instructions of one kind per quad, in four places
//                   (profiles/r05_issue_cost_probes.txt)
//   JPEZY_ABL_*     timing probes that give WRONG results (what a part of the kernel costs by leaving it out); such a build must also
//                   define JPEZY_EXPERIMENT (jpezy_experiment.h) and says so through jpezy_hip_is_experimental_build()
//   JPEZY_PS_*      build knobs of the persistent kernels
#pragma once

#ifdef JPEZY_PROBE_SALU   // timing probe (results unchanged): JPEZY_PROBE_SALU extra scalar-ALU instructions per quad, in four places
#define PROBE_SALU() do { int d_ = lane; d_ = __builtin_amdgcn_readfirstlane(d_); _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_SALU / 4; ++k_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(d_)); asm volatile("" :: "s"(d_)); } while (0)
#else
#define PROBE_SALU() do { } while (0)
#endif
#ifdef JPEZY_PROBE_VALU   // the same with full-rate vector instructions
#define PROBE_VALU() do { int d_ = lane; _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_VALU / 4; ++k_) asm volatile("v_add_u32 %0, %0, 1" : "+v"(d_)); asm volatile("" :: "v"(d_)); } while (0)
#else
#define PROBE_VALU() do { } while (0)
#endif
#ifdef JPEZY_PROBE_NOP    // s_nop 0
#define PROBE_NOP() do { _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_NOP / 4; ++k_) asm volatile("s_nop 0"); } while (0)
#else
#define PROBE_NOP() do { } while (0)
#endif
#ifdef JPEZY_PROBE_HALF   // a second-class vector instruction (v_cvt_f32_ubyte0)
#define PROBE_HALF() do { float d_ = __builtin_bit_cast(float, lane); _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_HALF / 4; ++k_) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(d_)); asm volatile("" :: "v"(d_)); } while (0)
#else
#define PROBE_HALF() do { } while (0)
#endif
#ifdef JPEZY_PROBE_PK     // a packed FP32 instruction
#define PROBE_PK() do { f2 d_ = { 1.f, 2.f }; _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_PK / 4; ++k_) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(d_)); asm volatile("" :: "v"(d_)); } while (0)
#else
#define PROBE_PK() do { } while (0)
#endif
#ifdef JPEZY_PROBE_LDS    // a 2-byte LDS store into the (still unused) queue area of the wave's slice
#define PROBE_LDS() do { _Pragma("unroll") for (int k_ = 0; k_ < JPEZY_PROBE_LDS / 4; ++k_) asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"((unsigned)(uintptr_t)(queue + 8) + 2u * (unsigned)lane), "v"(lane), "n"(0) : "memory"); } while (0)
#else
#define PROBE_LDS() do { } while (0)
#endif
#define PROBE_ALL() do { PROBE_SALU(); PROBE_VALU(); PROBE_NOP(); PROBE_HALF(); PROBE_PK(); PROBE_LDS(); } while (0)

// ---- JPEZY_ABL_*: parts of the kernel left out (wrong results; jpezy_experiment.h) ----
#ifdef JPEZY_ABL_NOCFLAG    // what the colour guard tests and their rare path cost
#define JPEZY_LAB_COLOUR_VOTE(x) (false)
#else
#define JPEZY_LAB_COLOUR_VOTE(x) (x)
#endif
#ifdef JPEZY_ABL_NOGUARD    // what the coefficient guard tests and levels 2/3 cost
#define JPEZY_LAB_GUARD_CAND(x) (false)
#else
#define JPEZY_LAB_GUARD_CAND(x) (x)
#endif
#ifdef JPEZY_ABL_NOSTORE    // the coefficients are staged and read back but never stored (only lanes whose staged data match a value they never have would store)
#define JPEZY_LAB_VALID_CHUNKS(x) ((lds[0] == 0x12345678u) ? 1 : 0)
#else
#define JPEZY_LAB_VALID_CHUNKS(x) (x)
#endif

// ---- build knobs of the persistent kernels (jpezy_kernels_f32_ps.hip) ----
#ifndef JPEZY_PS_CONSTS_LDS
#define JPEZY_PS_CONSTS_LDS 0  // 1: the persistent kernels read the lane's quantiser records per quad from an LDS copy of the tables instead of keeping them in 22 registers
#endif
#ifndef JPEZY_PS_DC_FORMULA
#define JPEZY_PS_DC_FORMULA 1  // 1: the persistent kernels compute the quantised DC (dc_formula, host-verified) instead of looking it up in a 32 KB table in LDS
#endif
#ifndef JPEZY_PS_DCQ_LDS
#define JPEZY_PS_DCQ_LDS (!JPEZY_PS_DC_FORMULA)   // without the formula: 1 = the quantised-DC tables copied to LDS, 0 = they stay in global memory (the loop then holds three byte loads)
#endif
#ifndef JPEZY_PS_HOOK_LATE
#define JPEZY_PS_HOOK_LATE 0   // 1: after_pixels() runs behind the luma quantiser (step 3+4) instead of behind step 2b: the next quad's pixel registers are not live across the most register-hungry phase
#endif
#ifndef JPEZY_PS_FENCES
#define JPEZY_PS_FENCES 1
#endif
