/* llx_debug.h - diagnostic entry points of libllx_hip.so (timing probes of the attention kernels).  NOT part of the drop-in
 * boundary (include/llx.h): nothing in llama-x_amd/ calls these; tools/attn_stamps.py and tools/attn_bwd_stamps.py do. */
#ifndef LLX_DEBUG_H
#define LLX_DEBUG_H

#include "llx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* in-kernel s_memtime stamps of the forward kernel (a separate stamp build; timing only, q/k/v contiguous [1,S,H,128]) */
int llx_debug_attn_fwd_stamps(const void* q, const void* k, const void* v, void* o, int64_t S, int64_t H, int64_t KVH,
                              unsigned long long* stamps, llx_stream_t s);
/* route the next llx_attn_bwd calls (causal, no flags) through the stamp build of the dK/dV kernel; NULL switches it off */
int llx_debug_attn_bwd_set_stamps(unsigned long long* stamps);
/* workgroups per CU the runtime grants the forward kernel (occupancy API; advisory) */
int llx_debug_attn_fwd_occupancy(void);

#ifdef __cplusplus
}
#endif
#endif
