// Hardware probe: checks the MFMA operand / accumulator lane maps and the
// ds_read_tr16_b64 gather that the llx kernels rely on, with exact integer data.
// Build:  hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o gpurun_out/probe
// Run on the GPU box; prints PASS/FAIL per hypothesis.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

static uint16_t f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

// ---- 16x16x32 bf16: A[16][32], B[32][16] -> C[16][16]
__global__ void k_mfma16(const uint16_t* A, const uint16_t* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    uint16_t av = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    uint16_t bv = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    a[j] = __builtin_bit_cast(__bf16, av);
    b[j] = __builtin_bit_cast(__bf16, bv);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// ---- 32x32x16 bf16: A[32][16], B[16][32] -> C[32][32]
__global__ void k_mfma32(const uint16_t* A, const uint16_t* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    uint16_t av = A[(l & 31) * 16 + 8 * (l >> 5) + j];
    uint16_t bv = B[(8 * (l >> 5) + j) * 32 + (l & 31)];
    a[j] = __builtin_bit_cast(__bf16, av);
    b[j] = __builtin_bit_cast(__bf16, bv);
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// ---- i8 32x32x32: A[32][32], B[32][32] -> C[32][32]; hypothesis k = 16*(l>>5)+j
__global__ void k_mfma32_i8(const int8_t* A, const int8_t* B, int* C) {
  int l = threadIdx.x;
  union { int8_t b[16]; i32x4 v; } a, b;
  for (int j = 0; j < 16; ++j) {
    a.b[j] = A[(l & 31) * 32 + 16 * (l >> 5) + j];
    b.b[j] = B[(16 * (l >> 5) + j) * 32 + (l & 31)];
  }
  i32x16 c = {};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// ---- i8 16x16x64: A[16][64], B[64][16]; hypothesis k = 16*(l>>4)+j
__global__ void k_mfma16_i8(const int8_t* A, const int8_t* B, int* C) {
  int l = threadIdx.x;
  union { int8_t b[16]; i32x4 v; } a, b;
  for (int j = 0; j < 16; ++j) {
    a.b[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
    b.b[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
  }
  i32x4 c = {};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// ---- ds_read_tr16_b64: LDS holds a [16 rows][64 cols] u16 image, value = row*64+col.
// Each 16-lane group g reads block rows 4g..4g+3, cols 0..15:
// lane 4q+p of the group supplies &img[4g+q][4p]; expect lane i gets img[4g+0..3][i].
__global__ void k_trread(int* out) {
  __shared__ __attribute__((aligned(16))) uint16_t img[16 * 64];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 64; i += 64) img[i] = (uint16_t)i;
  __syncthreads();
  int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  const uint16_t* addr = &img[(4 * g + q) * 64 + 4 * p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (int)(uint16_t)v[e];
}

// ---- accumulator-as-operand k-permutation for 32x32x16 (guide §3): X (32x32 f32 acc) -> bf16 B operand.
// Y = A.X with A[32][32] (rows i, k = X row), X[32][32]. For k-step s (0,1) lane half h element j
// corresponds to X row 16s + 8(j>>2) + 4h + (j&3).
__global__ void k_acc_operand(const uint16_t* A, const uint16_t* X1, const uint16_t* X2, float* Y) {
  // First compute X = X1 (32x16) * X2 (16x32) as an accumulator, then Y = A * X.
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = __builtin_bit_cast(__bf16, X1[(l & 31) * 16 + 8 * (l >> 5) + j]);
    b[j] = __builtin_bit_cast(__bf16, X2[(8 * (l >> 5) + j) * 32 + (l & 31)]);
  }
  f32x16 x = {};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 y = {};
  int h = l >> 5;
  for (int s = 0; s < 2; ++s) {
    bf16x8 xb, aa;
    for (int j = 0; j < 8; ++j) {
      xb[j] = (__bf16)x[8 * s + j];
      int krow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      aa[j] = __builtin_bit_cast(__bf16, A[(l & 31) * 32 + krow]);
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, xb, y, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) Y[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = y[r];
}

// ---- global_load_lds 16B: lane i lands at base + 16*i
__global__ void k_glds(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[64 * 4];
  int l = threadIdx.x;
  // lane l loads source chunk (63-l) -> expect LDS chunk l == src chunk 63-l
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * (63 - l)),
                                   (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = buf[l * 4 + e];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T> T* dev(const std::vector<T>& h) {
  T* d; hipMalloc(&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d;
}

int main() {
  int ok_all = 1;
  {  // 16x16x32
    std::vector<uint16_t> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16), C(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { Af[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5); A[i * 32 + k] = f2bf_host(Af[i * 32 + k]); }
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) { Bf[k * 16 + j] = (float)((k * 5 + j * 2 + k * j) % 13 - 6); B[k * 16 + j] = f2bf_host(Bf[k * 16 + j]); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += Af[i * 32 + k] * Bf[k * 16 + j];
    auto dA = dev(A); auto dB = dev(B); float* dC; hipMalloc(&dC, 256 * 4);
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 256; ++i) ok &= (C[i] == R[i]);
    printf("mfma_f32_16x16x32_bf16 layout: %s\n", ok ? "PASS" : "FAIL"); ok_all &= ok;
  }
  {  // 32x32x16
    std::vector<uint16_t> A(32 * 16), B(16 * 32); std::vector<float> Af(512), Bf(512), C(1024), R(1024, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) { Af[i * 16 + k] = (float)((i * 7 + k * 3) % 11 - 5); A[i * 16 + k] = f2bf_host(Af[i * 16 + k]); }
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) { Bf[k * 32 + j] = (float)((k * 5 + j * 2 + k * j) % 13 - 6); B[k * 32 + j] = f2bf_host(Bf[k * 32 + j]); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
    auto dA = dev(A); auto dB = dev(B); float* dC; hipMalloc(&dC, 1024 * 4);
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); hipMemcpy(C.data(), dC, 1024 * 4, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 1024; ++i) ok &= (C[i] == R[i]);
    printf("mfma_f32_32x32x16_bf16 layout: %s\n", ok ? "PASS" : "FAIL"); ok_all &= ok;
  }
  {  // i8 32x32x32
    std::vector<int8_t> A(1024), B(1024); std::vector<int> C(1024), R(1024, 0);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (int8_t)((i * 7 + k * 3) % 23 - 11);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (int8_t)((k * 5 + j * 2 + k * j) % 29 - 14);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 32; ++k) R[i * 32 + j] += (int)A[i * 32 + k] * (int)B[k * 32 + j];
    auto dA = dev(A); auto dB = dev(B); int* dC; hipMalloc(&dC, 4096);
    k_mfma32_i8<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 1024; ++i) ok &= (C[i] == R[i]);
    printf("mfma_i32_32x32x32_i8 layout (k=16h+j): %s\n", ok ? "PASS" : "FAIL"); ok_all &= ok;
  }
  {  // i8 16x16x64
    std::vector<int8_t> A(1024), B(1024); std::vector<int> C(256), R(256, 0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 64; ++k) A[i * 64 + k] = (int8_t)((i * 7 + k * 3) % 23 - 11);
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (int8_t)((k * 5 + j * 2 + k * j) % 29 - 14);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 64; ++k) R[i * 16 + j] += (int)A[i * 64 + k] * (int)B[k * 16 + j];
    auto dA = dev(A); auto dB = dev(B); int* dC; hipMalloc(&dC, 1024);
    k_mfma16_i8<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize()); hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 256; ++i) ok &= (C[i] == R[i]);
    printf("mfma_i32_16x16x64_i8 layout (k=16q+j): %s\n", ok ? "PASS" : "FAIL"); ok_all &= ok;
  }
  {  // tr read
    std::vector<int> out(256); int* d; hipMalloc(&d, 1024);
    k_trread<<<1, 64>>>(d); CK(hipDeviceSynchronize()); hipMemcpy(out.data(), d, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { int g = l >> 4, i = l & 15; ok &= (out[l * 4 + e] == (4 * g + e) * 64 + i); }
    printf("ds_read_tr16_b64 semantics: %s\n", ok ? "PASS" : "FAIL"); ok_all &= ok;
    if (!ok) for (int l = 0; l < 64; ++l) printf("  lane %d: %d %d %d %d\n", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
  }
  {  // acc as operand
    std::vector<uint16_t> A(1024), X1(512), X2(512); std::vector<float> Af(1024), X1f(512), X2f(512), X(1024, 0.f), Y(1024), R(1024, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 32; ++k) { Af[i * 32 + k] = (float)((i * 3 + k * 5 + i * k) % 7 - 3); A[i * 32 + k] = f2bf_host(Af[i * 32 + k]); }
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) { X1f[i * 16 + k] = (float)((i + 2 * k) % 5 - 2); X1[i * 16 + k] = f2bf_host(X1f[i * 16 + k]); }
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) { X2f[k * 32 + j] = (float)((3 * k + j + k * j) % 3 - 1); X2[k * 32 + j] = f2bf_host(X2f[k * 32 + j]); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) X[i * 32 + j] += X1f[i * 16 + k] * X2f[k * 32 + j];
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 32; ++k) R[i * 32 + j] += Af[i * 32 + k] * X[k * 32 + j];
    auto dA = dev(A); auto d1 = dev(X1); auto d2 = dev(X2); float* dY; hipMalloc(&dY, 4096);
    k_acc_operand<<<1, 64>>>(dA, d1, d2, dY); CK(hipDeviceSynchronize()); hipMemcpy(Y.data(), dY, 4096, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 1024; ++i) ok &= (Y[i] == R[i]);
    printf("acc-as-B-operand k permutation: %s\n", ok ? "PASS" : "FAIL"); ok_all &= ok;
  }
  {  // glds
    std::vector<uint32_t> src(256), out(256); for (int i = 0; i < 256; ++i) src[i] = 1000 + i;
    auto ds = dev(src); uint32_t* d; hipMalloc(&d, 1024);
    k_glds<<<1, 64>>>(ds, d); CK(hipDeviceSynchronize()); hipMemcpy(out.data(), d, 1024, hipMemcpyDeviceToHost);
    int ok = 1; for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) ok &= (out[l * 4 + e] == 1000u + 4 * (63 - l) + e);
    printf("global_load_lds x16 lane-linear dest: %s\n", ok ? "PASS" : "FAIL"); ok_all &= ok;
  }
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  printf("device: %s CUs=%d clock=%d kHz lds/block=%zu\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock);
  printf("PROBE %s\n", ok_all ? "ALL PASS" : "SOME FAIL");
  return ok_all ? 0 : 2;
}
