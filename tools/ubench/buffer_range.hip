// buffer_range.hip -- how does gfx950 range-check a raw buffer load (stride 0) whose access STARTS inside num_records but ENDS
// past it?  The triple-column form of the column-wave kernel streams pixel triples as unaligned dword loads through a buffer
// resource whose num_records is the row width; if only the start of an access is checked, the dword of a position near the
// row's end reads up to 3 bytes past the row -- on the last row of the last frame, past the caller's allocation.
// The buffer here is 256 bytes of 0x11, 0x12, ... inside a 4 KiB allocation (reading past num_records stays inside it).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned char* base, int nrec, unsigned* out)
{
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, nrec, 0x00020000);
  const int off = nrec - 8 + (int)threadIdx.x;                 // offsets nrec-8 .. nrec+7
  out[threadIdx.x] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
  out[16 + threadIdx.x] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs, off, 0, 0);
  out[32 + threadIdx.x] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, 0);
}
int main()
{
  unsigned char h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (unsigned char)(0x11 + (i & 0x7f));
  unsigned char* d; unsigned* o; unsigned ho[48];
  if (hipMalloc((void**)&d, 4096) != hipSuccess || hipMalloc((void**)&o, sizeof(ho)) != hipSuccess) return 1;
  (void)hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  for (int nrec : {256, 253}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(16), 0, 0, d, nrec, o);
    if (hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("num_records = %d (bytes at offset i hold 0x11 + (i & 0x7f))\n", nrec);
    for (int t = 0; t < 16; ++t) {
      const int off = nrec - 8 + t;
      printf("  offset %4d (%+d from the end): dword %08x   byte %02x   u16 %04x   -> %s\n", off, off - nrec, ho[t], ho[16 + t], ho[32 + t],
             off + 4 <= nrec ? "whole dword inside" : off < nrec ? "dword STARTS inside, ENDS past num_records" : "starts past num_records");
    }
  }
  return 0;
}
