// Global -> LDS copies (LDS-DMA, global_load_lds_dwordx4: 1 KiB per wave instruction, no staging registers) that hipcc does
// NOT count.
//
// Why not __builtin_amdgcn_global_load_lds: hipcc (ROCm 7.2) treats an LDS-DMA as a pending LDS write on the VM counter and,
// when it cannot prove that a later ds_read touches a different buffer (ring slot chosen at run time: `ci % RING`), inserts
// `s_waitcnt vmcnt(0)` in front of the first ds_read behind the copy.  In the wgrad kernels that drained the whole operand
// ring once per chunk -- the copies of chunk c+1.. were waited for before chunk c was even read -- so the "RING - 1 chunks in
// flight" the kernels were designed around never existed (seen in the ISA: wgrad.hip / wgrad_bf16.hip of round 2 carry a
// vmcnt(0) between the stage issue and the operand reads; cdna_hip_programming.md "HIP compiler defeats it").  An asm copy is
// invisible to that bookkeeping; the kernels order the data themselves:
//     issuing wave: s_waitcnt vmcnt(N) counting its own later copies  ->  s_barrier  ->  ds_read by any wave.
// M0 carries the LDS destination (wave-uniform byte address; the hardware adds lane * 16); it is compiler-reserved, so it is
// saved and restored inside the statement that writes it (cdna_hip_programming.md section 5, "LDS-DMA recipe").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrf {

// LDS byte address (the M0 value) of a pointer into the workgroup's dynamic LDS, as a wave-uniform scalar
__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
  typedef __attribute__((address_space(3))) const char lds_cchar;
  return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_cchar*)p);
}

// 16 bytes per lane from `gsrc` (per-lane address) to LDS byte address lds_dst + lane * 16.  NT: non-temporal (streamed once).
template <bool NT>
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// The scalar-base form: 16 bytes per lane from (wave-uniform base) + voff (VGPR, bytes) to LDS byte address lds_dst + lane * 16.  A
// copy costs the issuing wave no per-lane 64-bit address arithmetic: the base walks in SGPRs.
template <bool NT>
__device__ __forceinline__ void lds_dma16s(const char* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace nrf
