// pt_arith.hip -- the arithmetic-bearing kernels of pt_engine.hip once more, as a code object of another arithmetic level.
//
// Compiled with -DPT_ARITH=1 (CONTRACTED: -ffp-contract=fast) and with -DPT_ARITH=2 (FAST: also
// -fno-hip-fp32-correctly-rounded-divide-sqrt); see pt_device.h for what each level means and include/mi355x_pathtracer.h
// (ptx_options.arith) for how a caller asks for one.  The default level 0 -- exact, bit-identical to the CPU oracle, the level every
// headline number and every parity claim is made at -- is pt_engine.hip compiled by itself.
//
// What this file adds to the library is ONE symbol, ptx_arith_kernels_<level>(): a table of launchers for k_bounce / k_mesh /
// k_finish and the per-stage test kernels as compiled here.  Host code, scene upload, buffers, launch plans, gather, preview: all of
// it stays the level-0 translation unit's (pt_engine.hip under `#if PT_ARITH == 0`); the tables the host computes with the device's
// own functions (tabulated normals) are therefore the exact ones at every level.
//
// `ptd` is renamed for this translation unit: every inline function of pt_device.h / pt_bvh.h compiled with these flags gets a
// symbol of its own, so the linker can never hand the exact translation unit a contracted copy (or the other way round).
#ifndef PT_ARITH
#error "pt_arith.hip is compiled with -DPT_ARITH=1 or -DPT_ARITH=2"
#endif
#if PT_ARITH == 1
#define ptd ptd_arith1
#elif PT_ARITH == 2
#define ptd ptd_arith2
#else
#error "PT_ARITH must be 1 or 2 here"
#endif
#include "pt_engine.hip"
