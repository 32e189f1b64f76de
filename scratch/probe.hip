// Probe: ds_read_tr16_b64 semantics + MFMA fragment layouts on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void k_tr(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x;
  auto v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4_t __attribute__((address_space(3)))*)(&lds[l * 4]));
  unsigned short* pv = (unsigned short*)&v;
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = pv[j];
}

static __device__ __bf16 tobf(float x) { return (__bf16)x; }

// C[16x16] = A[16x32] * B[32x16], assumed layouts
__global__ void k_mfma16(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = tobf(A[(l & 15) * 32 + (l >> 4) * 8 + j]);
    b[j] = tobf(B[((l >> 4) * 8 + j) * 16 + (l & 15)]);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// C[32x32] = A[32x16] * B[16x32]
__global__ void k_mfma32(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = tobf(A[(l & 31) * 16 + (l >> 5) * 8 + j]);
    b[j] = tobf(B[((l >> 5) * 8 + j) * 32 + (l & 31)]);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// f32: C[32x32] = A[32x2]*B[2x32]
__global__ void k_mfma32f(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)];
  float b = B[(l >> 5) * 32 + (l & 31)];
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// f32: C[16x16] = A[16x4]*B[4x16]
__global__ void k_mfma16f(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)];
  float b = B[(l >> 4) * 16 + (l & 15)];
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

static void check(const char* name, int M, int N, int K, void (*launch)(const float*, const float*, float*)) {
  std::vector<float> A(M * K), B(K * N), C(M * N), R(M * N, 0.f);
  for (auto& x : A) x = (float)((rand() % 7) - 3);
  for (auto& x : B) x = (float)((rand() % 5) - 2);
  for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { float s = 0; for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * N + j]; R[i * N + j] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  launch(dA, dB, dC);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < M * N; ++i) bad += (C[i] != R[i]);
  printf("%s: mismatches %d / %d\n", name, bad, M * N);
}

int main() {
  unsigned short* d; hipMalloc(&d, 256 * 2);
  k_tr<<<1, 64>>>(d);
  unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("tr16_b64: lane: 4 values (value = src_lane*4+src_elem)\n");
  for (int l = 0; l < 64; ++l) printf("L%02d: %3d %3d %3d %3d   [src lanes %d %d %d %d | elems %d %d %d %d]\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], h[l*4]/4, h[l*4+1]/4, h[l*4+2]/4, h[l*4+3]/4, h[l*4]%4, h[l*4+1]%4, h[l*4+2]%4, h[l*4+3]%4);
  check("mfma16x16x32_bf16", 16, 16, 32, [](const float* a, const float* b, float* c) { k_mfma16<<<1, 64>>>(a, b, c); });
  check("mfma32x32x16_bf16", 32, 32, 16, [](const float* a, const float* b, float* c) { k_mfma32<<<1, 64>>>(a, b, c); });
  check("mfma32x32x2_f32", 32, 32, 2, [](const float* a, const float* b, float* c) { k_mfma32f<<<1, 64>>>(a, b, c); });
  check("mfma16x16x4_f32", 16, 16, 4, [](const float* a, const float* b, float* c) { k_mfma16f<<<1, 64>>>(a, b, c); });
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("dev %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  return 0;
}
