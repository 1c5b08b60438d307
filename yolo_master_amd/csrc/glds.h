// LDS-DMA helpers shared by the kernels that stage operands with `buffer_load_dwordx4 ... lds` (csrc/conv_glds.hip; tools/micro/parked/esfused.hip.txt used them too):
// raw buffer resources, the load itself, and the wait / fence idioms around a workgroup barrier.  tests/hostemu provides plain-pointer
// stand-ins (the transfer completes at once there, so counted-vmcnt mistakes are invisible on the emulator; addressing and masks are not).
#pragma once
#include "ymk_common.h"

#ifndef YMK_HOST_EMU
typedef __attribute__((address_space(1))) const void* glds_gptr;
typedef __attribute__((address_space(3))) void* glds_lptr;
typedef __amdgpu_buffer_rsrc_t glds_rsrc;
// raw buffer (stride 0) of n bytes at p; word 3 = 32-bit raw data format (what the dword loads need on gfx9-class hardware)
#define GLDS_MAKE_RSRC(p, n) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(p)), 0, (int)(n), 0x00020000)
// buffer_load_dwordx4 ... offen lds: lane -> 16 bytes from base + voff (VGPR) + soff (SGPR) to the LDS base in M0 + lane * 16; lanes
// with voff + 16 > num_records - soff get zeros and touch no memory (hardware range check)
#define GLDS_BUFFER_LOAD_LDS(rs, dst, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (glds_lptr)(dst), 16, (int)(voff), (int)(soff), 0, 0)
#define GLDS_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define GLDS_COMPILER_FENCE() asm volatile("" ::: "memory")
#else   // tests/hostemu: plain pointers, no inline assembly
typedef const void* glds_gptr;
typedef void* glds_lptr;
typedef hostemu_rsrc glds_rsrc;
#define GLDS_MAKE_RSRC(p, n) hostemu_make_rsrc(p, (unsigned)(n))
#define GLDS_BUFFER_LOAD_LDS(rs, dst, voff, soff) hostemu_buffer_load_lds16(rs, (void*)(dst), (unsigned)(voff), (unsigned)(soff))
#define GLDS_WAIT_LGKM0() ((void)0)
#define GLDS_COMPILER_FENCE() ((void)0)
#endif

// s_waitcnt immediate (gfx9 family): vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]
#define GLDS_WAITCNT_VM(n) (0x0F70 | ((n) & 15) | ((((n) >> 4) & 3) << 14))
