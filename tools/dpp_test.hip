// Checks the lane semantics of fused VOP2+DPP wave shifts on gfx950 against v_mov_b32_dpp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define T(NAME, ASM) \
__global__ void NAME(const unsigned* a, const unsigned* b, unsigned* o){ unsigned x=a[threadIdx.x], y=b[threadIdx.x], r=0xdeadbeef; \
  asm volatile("s_nop 4\n" ASM "\ns_nop 4" : "+v"(r) : "v"(x), "v"(y)); o[threadIdx.x]=r; }
T(k_mov_shr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_mov_shl, "v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_add_shr, "v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_add_shl, "v_add_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_subrev_shr, "v_subrev_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_subrev_shl, "v_subrev_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_sub_shr, "v_sub_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_sub_shl, "v_sub_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_add_rowshr, "v_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
T(k_subrev_rowshr, "v_subrev_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
typedef void(*K)(const unsigned*, const unsigned*, unsigned*);
int main(){
  unsigned ha[64], hb[64], ho[64]; for(int i=0;i<64;i++){ha[i]=1000+i; hb[i]=100000+100*i;}
  unsigned *a,*b,*o; hipMalloc(&a,256); hipMalloc(&b,256); hipMalloc(&o,256);
  hipMemcpy(a,ha,256,hipMemcpyHostToDevice); hipMemcpy(b,hb,256,hipMemcpyHostToDevice);
  struct {const char* n; K k; int shift; int op;} ts[] = {
   {"mov_shr",k_mov_shr,-1,0},{"mov_shl",k_mov_shl,1,0},{"add_shr",k_add_shr,-1,1},{"add_shl",k_add_shl,1,1},
   {"subrev_shr",k_subrev_shr,-1,2},{"subrev_shl",k_subrev_shl,1,2},{"sub_shr",k_sub_shr,-1,3},{"sub_shl",k_sub_shl,1,3},
   {"add_rowshr",k_add_rowshr,-1,1},{"subrev_rowshr",k_subrev_rowshr,-1,2}};
  for(auto& t: ts){
    hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, a, b, o); hipMemcpy(ho,o,256,hipMemcpyDeviceToHost);
    int bad=0; int first=-1;
    for(int l=0;l<64;l++){ int s=l+t.shift; bool rowedge = (t.n[4]=='r'&&t.n[7]=='s') ; 
      unsigned src = (s<0||s>63)?0u:ha[s]; 
      if(strstr(t.n,"rowshr") && (l%16==0)) src=0;
      unsigned e = t.op==0?src : t.op==1? src+hb[l] : t.op==2? hb[l]-src : src-hb[l];
      if(ho[l]!=e){bad++; if(first<0)first=l;} }
    printf("%-14s bad=%d first=%d  o[0]=%u o[1]=%u o[16]=%u o[17]=%u o[63]=%u\n", t.n,bad,first,ho[0],ho[1],ho[16],ho[17],ho[63]);
  }
  return 0;
}
