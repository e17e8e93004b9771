// vgpr_bank.hip -- does a VALU op whose two VGPR sources share a register bank (index mod 4) issue slower on gfx950?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

#define BODY_SAME \
    "v_bcnt_u32_b32 v8, v16, v8\n v_bcnt_u32_b32 v9, v17, v9\n v_bcnt_u32_b32 v10, v18, v10\n v_bcnt_u32_b32 v11, v19, v11\n" \
    "v_bcnt_u32_b32 v12, v20, v12\n v_bcnt_u32_b32 v13, v21, v13\n v_bcnt_u32_b32 v14, v22, v14\n v_bcnt_u32_b32 v15, v23, v15\n"
#define BODY_DIFF \
    "v_bcnt_u32_b32 v8, v17, v8\n v_bcnt_u32_b32 v9, v18, v9\n v_bcnt_u32_b32 v10, v19, v10\n v_bcnt_u32_b32 v11, v20, v11\n" \
    "v_bcnt_u32_b32 v12, v21, v12\n v_bcnt_u32_b32 v13, v22, v13\n v_bcnt_u32_b32 v14, v23, v14\n v_bcnt_u32_b32 v15, v16, v15\n"
#define XBODY_SAME \
    "v_xor_b32 v8, v16, v8\n v_xor_b32 v9, v17, v9\n v_xor_b32 v10, v18, v10\n v_xor_b32 v11, v19, v11\n" \
    "v_xor_b32 v12, v20, v12\n v_xor_b32 v13, v21, v13\n v_xor_b32 v14, v22, v14\n v_xor_b32 v15, v23, v15\n"
#define XBODY_DIFF \
    "v_xor_b32 v8, v17, v8\n v_xor_b32 v9, v18, v9\n v_xor_b32 v10, v19, v10\n v_xor_b32 v11, v20, v11\n" \
    "v_xor_b32 v12, v21, v12\n v_xor_b32 v13, v22, v13\n v_xor_b32 v14, v23, v14\n v_xor_b32 v15, v16, v15\n"
#define MBODY_SAME \
    "v_mul_f32 v8, v16, v8\n v_mul_f32 v9, v17, v9\n v_mul_f32 v10, v18, v10\n v_mul_f32 v11, v19, v11\n" \
    "v_mul_f32 v12, v20, v12\n v_mul_f32 v13, v21, v13\n v_mul_f32 v14, v22, v14\n v_mul_f32 v15, v23, v15\n"
#define MBODY_DIFF \
    "v_mul_f32 v8, v17, v8\n v_mul_f32 v9, v18, v9\n v_mul_f32 v10, v19, v10\n v_mul_f32 v11, v20, v11\n" \
    "v_mul_f32 v12, v21, v12\n v_mul_f32 v13, v22, v13\n v_mul_f32 v14, v23, v14\n v_mul_f32 v15, v16, v15\n"
#define PBODY_SAME \
    "v_pk_mul_f32 v[8:9], v[16:17], v[8:9]\n v_pk_mul_f32 v[10:11], v[18:19], v[10:11]\n v_pk_mul_f32 v[12:13], v[20:21], v[12:13]\n v_pk_mul_f32 v[14:15], v[22:23], v[14:15]\n"
#define PBODY_DIFF \
    "v_pk_mul_f32 v[8:9], v[18:19], v[8:9]\n v_pk_mul_f32 v[10:11], v[20:21], v[10:11]\n v_pk_mul_f32 v[12:13], v[22:23], v[12:13]\n v_pk_mul_f32 v[14:15], v[16:17], v[14:15]\n"


#define HBODY_SGPR \
    "v_xor_b32 v24, s21, v16\n v_bcnt_u32_b32 v8, v24, v8\n v_xor_b32 v25, s22, v17\n v_bcnt_u32_b32 v9, v25, v9\n" \
    "v_xor_b32 v26, s23, v18\n v_bcnt_u32_b32 v10, v26, v10\n v_xor_b32 v27, s24, v19\n v_bcnt_u32_b32 v11, v27, v11\n" \
    "v_xor_b32 v28, s21, v20\n v_bcnt_u32_b32 v12, v28, v12\n v_xor_b32 v29, s22, v21\n v_bcnt_u32_b32 v13, v29, v13\n" \
    "v_xor_b32 v30, s23, v22\n v_bcnt_u32_b32 v14, v30, v14\n v_xor_b32 v31, s24, v23\n v_bcnt_u32_b32 v15, v31, v15\n"
#define HBODY_VGPR \
    "v_xor_b32 v24, v32, v16\n v_bcnt_u32_b32 v8, v24, v8\n v_xor_b32 v25, v33, v17\n v_bcnt_u32_b32 v9, v25, v9\n" \
    "v_xor_b32 v26, v34, v18\n v_bcnt_u32_b32 v10, v26, v10\n v_xor_b32 v27, v35, v19\n v_bcnt_u32_b32 v11, v27, v11\n" \
    "v_xor_b32 v28, v32, v20\n v_bcnt_u32_b32 v12, v28, v12\n v_xor_b32 v29, v33, v21\n v_bcnt_u32_b32 v13, v29, v13\n" \
    "v_xor_b32 v30, v34, v22\n v_bcnt_u32_b32 v14, v30, v14\n v_xor_b32 v31, v35, v23\n v_bcnt_u32_b32 v15, v31, v15\n"
#define HBODY_INPLACE \
    "v_xor_b32 v16, s21, v16\n v_bcnt_u32_b32 v8, v16, v8\n v_xor_b32 v17, s22, v17\n v_bcnt_u32_b32 v9, v17, v9\n" \
    "v_xor_b32 v18, s23, v18\n v_bcnt_u32_b32 v10, v18, v10\n v_xor_b32 v19, s24, v19\n v_bcnt_u32_b32 v11, v19, v11\n" \
    "v_xor_b32 v20, s21, v20\n v_bcnt_u32_b32 v12, v20, v12\n v_xor_b32 v21, s22, v21\n v_bcnt_u32_b32 v13, v21, v13\n" \
    "v_xor_b32 v22, s23, v22\n v_bcnt_u32_b32 v14, v22, v14\n v_xor_b32 v23, s24, v23\n v_bcnt_u32_b32 v15, v23, v15\n"
#define HBODY_XONLY \
    "v_xor_b32 v24, s21, v16\n v_xor_b32 v25, s22, v17\n v_xor_b32 v26, s23, v18\n v_xor_b32 v27, s24, v19\n" \
    "v_xor_b32 v28, s21, v20\n v_xor_b32 v29, s22, v21\n v_xor_b32 v30, s23, v22\n v_xor_b32 v31, s24, v23\n"
#define HBODY_BONLY \
    "v_bcnt_u32_b32 v8, v24, v8\n v_bcnt_u32_b32 v9, v25, v9\n v_bcnt_u32_b32 v10, v26, v10\n v_bcnt_u32_b32 v11, v27, v11\n" \
    "v_bcnt_u32_b32 v12, v28, v12\n v_bcnt_u32_b32 v13, v29, v13\n v_bcnt_u32_b32 v14, v30, v14\n v_bcnt_u32_b32 v15, v31, v15\n"

#define HBODY_IPV \
    "v_xor_b32 v16, v32, v16\n v_bcnt_u32_b32 v8, v16, v8\n v_xor_b32 v17, v33, v17\n v_bcnt_u32_b32 v9, v17, v9\n" \
    "v_xor_b32 v18, v34, v18\n v_bcnt_u32_b32 v10, v18, v10\n v_xor_b32 v19, v35, v19\n v_bcnt_u32_b32 v11, v19, v11\n" \
    "v_xor_b32 v20, v32, v20\n v_bcnt_u32_b32 v12, v20, v12\n v_xor_b32 v21, v33, v21\n v_bcnt_u32_b32 v13, v21, v13\n" \
    "v_xor_b32 v22, v34, v22\n v_bcnt_u32_b32 v14, v22, v14\n v_xor_b32 v23, v35, v23\n v_bcnt_u32_b32 v15, v23, v15\n"
#define HBODY_IPV_BATCH \
    "v_xor_b32 v16, v32, v16\n v_xor_b32 v17, v33, v17\n v_xor_b32 v18, v34, v18\n v_xor_b32 v19, v35, v19\n" \
    "v_xor_b32 v20, v32, v20\n v_xor_b32 v21, v33, v21\n v_xor_b32 v22, v34, v22\n v_xor_b32 v23, v35, v23\n" \
    "v_bcnt_u32_b32 v8, v16, v8\n v_bcnt_u32_b32 v9, v17, v9\n v_bcnt_u32_b32 v10, v18, v10\n v_bcnt_u32_b32 v11, v19, v11\n" \
    "v_bcnt_u32_b32 v12, v20, v12\n v_bcnt_u32_b32 v13, v21, v13\n v_bcnt_u32_b32 v14, v22, v14\n v_bcnt_u32_b32 v15, v23, v15\n"
#define HBODY_IPV_SAMEACC \
    "v_xor_b32 v16, v32, v16\n v_bcnt_u32_b32 v16, v33, v16\n v_xor_b32 v17, v33, v17\n v_bcnt_u32_b32 v17, v33, v17\n" \
    "v_xor_b32 v18, v34, v18\n v_bcnt_u32_b32 v18, v33, v18\n v_xor_b32 v19, v35, v19\n v_bcnt_u32_b32 v19, v33, v19\n" \
    "v_xor_b32 v20, v32, v20\n v_bcnt_u32_b32 v20, v33, v20\n v_xor_b32 v21, v33, v21\n v_bcnt_u32_b32 v21, v33, v21\n" \
    "v_xor_b32 v22, v34, v22\n v_bcnt_u32_b32 v22, v33, v22\n v_xor_b32 v23, v35, v23\n v_bcnt_u32_b32 v23, v33, v23\n"
#define HBODY_IPV_XONLY \
    "v_xor_b32 v16, v32, v16\n v_xor_b32 v17, v33, v17\n v_xor_b32 v18, v34, v18\n v_xor_b32 v19, v35, v19\n" \
    "v_xor_b32 v20, v32, v20\n v_xor_b32 v21, v33, v21\n v_xor_b32 v22, v34, v22\n v_xor_b32 v23, v35, v23\n"

#define HBODY_S0 \
    "v_xor_b32 v16, v16, v32\n v_bcnt_u32_b32 v8, v16, v8\n v_xor_b32 v17, v17, v33\n v_bcnt_u32_b32 v9, v17, v9\n" \
    "v_xor_b32 v18, v18, v34\n v_bcnt_u32_b32 v10, v18, v10\n v_xor_b32 v19, v19, v35\n v_bcnt_u32_b32 v11, v19, v11\n" \
    "v_xor_b32 v20, v20, v32\n v_bcnt_u32_b32 v12, v20, v12\n v_xor_b32 v21, v21, v33\n v_bcnt_u32_b32 v13, v21, v13\n" \
    "v_xor_b32 v22, v22, v34\n v_bcnt_u32_b32 v14, v22, v14\n v_xor_b32 v23, v23, v35\n v_bcnt_u32_b32 v15, v23, v15\n"
#define HBODY_S0_ONE \
    "v_xor_b32 v16, v16, v32\n v_bcnt_u32_b32 v16, v32, v16\n v_xor_b32 v17, v17, v32\n v_bcnt_u32_b32 v17, v32, v17\n" \
    "v_xor_b32 v18, v18, v32\n v_bcnt_u32_b32 v18, v32, v18\n v_xor_b32 v19, v19, v32\n v_bcnt_u32_b32 v19, v32, v19\n" \
    "v_xor_b32 v20, v20, v32\n v_bcnt_u32_b32 v20, v32, v20\n v_xor_b32 v21, v21, v32\n v_bcnt_u32_b32 v21, v32, v21\n" \
    "v_xor_b32 v22, v22, v32\n v_bcnt_u32_b32 v22, v32, v22\n v_xor_b32 v23, v23, v32\n v_bcnt_u32_b32 v23, v32, v23\n"

#define NB_X "v_xor_b32 v16, v32, v16\n s_nop 0\n v_bcnt_u32_b32 v8, v16, v8\n v_xor_b32 v17, v33, v17\n s_nop 0\n v_bcnt_u32_b32 v9, v17, v9\n v_xor_b32 v18, v34, v18\n s_nop 0\n v_bcnt_u32_b32 v10, v18, v10\n v_xor_b32 v19, v35, v19\n s_nop 0\n v_bcnt_u32_b32 v11, v19, v11\n v_xor_b32 v20, v32, v20\n s_nop 0\n v_bcnt_u32_b32 v12, v20, v12\n v_xor_b32 v21, v33, v21\n s_nop 0\n v_bcnt_u32_b32 v13, v21, v13\n v_xor_b32 v22, v34, v22\n s_nop 0\n v_bcnt_u32_b32 v14, v22, v14\n v_xor_b32 v23, v35, v23\n s_nop 0\n v_bcnt_u32_b32 v15, v23, v15\n"
#define NB_XB "v_xor_b32 v16, v32, v16\n s_nop 0\n v_bcnt_u32_b32 v8, v16, v8\n s_nop 0\n v_xor_b32 v17, v33, v17\n s_nop 0\n v_bcnt_u32_b32 v9, v17, v9\n s_nop 0\n v_xor_b32 v18, v34, v18\n s_nop 0\n v_bcnt_u32_b32 v10, v18, v10\n s_nop 0\n v_xor_b32 v19, v35, v19\n s_nop 0\n v_bcnt_u32_b32 v11, v19, v11\n s_nop 0\n v_xor_b32 v20, v32, v20\n s_nop 0\n v_bcnt_u32_b32 v12, v20, v12\n s_nop 0\n v_xor_b32 v21, v33, v21\n s_nop 0\n v_bcnt_u32_b32 v13, v21, v13\n s_nop 0\n v_xor_b32 v22, v34, v22\n s_nop 0\n v_bcnt_u32_b32 v14, v22, v14\n s_nop 0\n v_xor_b32 v23, v35, v23\n s_nop 0\n v_bcnt_u32_b32 v15, v23, v15\n s_nop 0\n"
#define NB_B "v_xor_b32 v16, v32, v16\n v_bcnt_u32_b32 v8, v16, v8\n s_nop 0\n v_xor_b32 v17, v33, v17\n v_bcnt_u32_b32 v9, v17, v9\n s_nop 0\n v_xor_b32 v18, v34, v18\n v_bcnt_u32_b32 v10, v18, v10\n s_nop 0\n v_xor_b32 v19, v35, v19\n v_bcnt_u32_b32 v11, v19, v11\n s_nop 0\n v_xor_b32 v20, v32, v20\n v_bcnt_u32_b32 v12, v20, v12\n s_nop 0\n v_xor_b32 v21, v33, v21\n v_bcnt_u32_b32 v13, v21, v13\n s_nop 0\n v_xor_b32 v22, v34, v22\n v_bcnt_u32_b32 v14, v22, v14\n s_nop 0\n v_xor_b32 v23, v35, v23\n v_bcnt_u32_b32 v15, v23, v15\n s_nop 0\n"
#define NB_SX "v_xor_b32 v24, s21, v16\n s_nop 0\n v_bcnt_u32_b32 v8, v24, v8\n v_xor_b32 v25, s22, v17\n s_nop 0\n v_bcnt_u32_b32 v9, v25, v9\n v_xor_b32 v26, s23, v18\n s_nop 0\n v_bcnt_u32_b32 v10, v26, v10\n v_xor_b32 v27, s24, v19\n s_nop 0\n v_bcnt_u32_b32 v11, v27, v11\n v_xor_b32 v28, s21, v20\n s_nop 0\n v_bcnt_u32_b32 v12, v28, v12\n v_xor_b32 v29, s22, v21\n s_nop 0\n v_bcnt_u32_b32 v13, v29, v13\n v_xor_b32 v30, s23, v22\n s_nop 0\n v_bcnt_u32_b32 v14, v30, v14\n v_xor_b32 v31, s24, v23\n s_nop 0\n v_bcnt_u32_b32 v15, v31, v15\n"
#define NB_SXB "v_xor_b32 v24, s21, v16\n s_nop 0\n v_bcnt_u32_b32 v8, v24, v8\n s_nop 0\n v_xor_b32 v25, s22, v17\n s_nop 0\n v_bcnt_u32_b32 v9, v25, v9\n s_nop 0\n v_xor_b32 v26, s23, v18\n s_nop 0\n v_bcnt_u32_b32 v10, v26, v10\n s_nop 0\n v_xor_b32 v27, s24, v19\n s_nop 0\n v_bcnt_u32_b32 v11, v27, v11\n s_nop 0\n v_xor_b32 v28, s21, v20\n s_nop 0\n v_bcnt_u32_b32 v12, v28, v12\n s_nop 0\n v_xor_b32 v29, s22, v21\n s_nop 0\n v_bcnt_u32_b32 v13, v29, v13\n s_nop 0\n v_xor_b32 v30, s23, v22\n s_nop 0\n v_bcnt_u32_b32 v14, v30, v14\n s_nop 0\n v_xor_b32 v31, s24, v23\n s_nop 0\n v_bcnt_u32_b32 v15, v31, v15\n s_nop 0\n"

#define LBODY "v_xor_b32 v40, v24, v40\n v_xor_b32 v48, v24, v48\n v_xor_b32 v56, v24, v56\n v_xor_b32 v64, v24, v64\n v_xor_b32 v41, v25, v41\n v_xor_b32 v49, v25, v49\n v_xor_b32 v57, v25, v57\n v_xor_b32 v65, v25, v65\n v_xor_b32 v42, v26, v42\n v_xor_b32 v50, v26, v50\n v_xor_b32 v58, v26, v58\n v_xor_b32 v66, v26, v66\n v_xor_b32 v43, v27, v43\n v_xor_b32 v51, v27, v51\n v_xor_b32 v59, v27, v59\n v_xor_b32 v67, v27, v67\n v_xor_b32 v44, v28, v44\n v_xor_b32 v52, v28, v52\n v_xor_b32 v60, v28, v60\n v_xor_b32 v68, v28, v68\n v_xor_b32 v45, v29, v45\n v_xor_b32 v53, v29, v53\n v_xor_b32 v61, v29, v61\n v_xor_b32 v69, v29, v69\n v_xor_b32 v46, v30, v46\n v_xor_b32 v54, v30, v54\n v_xor_b32 v62, v30, v62\n v_xor_b32 v70, v30, v70\n v_xor_b32 v47, v31, v47\n v_xor_b32 v55, v31, v55\n v_xor_b32 v63, v31, v63\n v_xor_b32 v71, v31, v71\n v_bcnt_u32_b32 v8, v40, v8\n v_bcnt_u32_b32 v9, v48, v9\n v_bcnt_u32_b32 v10, v56, v10\n v_bcnt_u32_b32 v11, v64, v11\n v_bcnt_u32_b32 v8, v41, v8\n v_bcnt_u32_b32 v9, v49, v9\n v_bcnt_u32_b32 v10, v57, v10\n v_bcnt_u32_b32 v11, v65, v11\n v_bcnt_u32_b32 v8, v42, v8\n v_bcnt_u32_b32 v9, v50, v9\n v_bcnt_u32_b32 v10, v58, v10\n v_bcnt_u32_b32 v11, v66, v11\n v_bcnt_u32_b32 v8, v43, v8\n v_bcnt_u32_b32 v9, v51, v9\n v_bcnt_u32_b32 v10, v59, v10\n v_bcnt_u32_b32 v11, v67, v11\n v_bcnt_u32_b32 v8, v44, v8\n v_bcnt_u32_b32 v9, v52, v9\n v_bcnt_u32_b32 v10, v60, v10\n v_bcnt_u32_b32 v11, v68, v11\n v_bcnt_u32_b32 v8, v45, v8\n v_bcnt_u32_b32 v9, v53, v9\n v_bcnt_u32_b32 v10, v61, v10\n v_bcnt_u32_b32 v11, v69, v11\n v_bcnt_u32_b32 v8, v46, v8\n v_bcnt_u32_b32 v9, v54, v9\n v_bcnt_u32_b32 v10, v62, v10\n v_bcnt_u32_b32 v11, v70, v11\n v_bcnt_u32_b32 v8, v47, v8\n v_bcnt_u32_b32 v9, v55, v9\n v_bcnt_u32_b32 v10, v63, v10\n v_bcnt_u32_b32 v11, v71, v11\n"
#define LBODY2 "v_xor_b32 v40, v24, v40\n v_xor_b32 v48, v24, v48\n v_xor_b32 v56, v24, v56\n v_xor_b32 v64, v24, v64\n v_bcnt_u32_b32 v8, v40, v8\n v_bcnt_u32_b32 v9, v48, v9\n v_bcnt_u32_b32 v10, v56, v10\n v_bcnt_u32_b32 v11, v64, v11\n v_xor_b32 v41, v25, v41\n v_xor_b32 v49, v25, v49\n v_xor_b32 v57, v25, v57\n v_xor_b32 v65, v25, v65\n v_bcnt_u32_b32 v8, v41, v8\n v_bcnt_u32_b32 v9, v49, v9\n v_bcnt_u32_b32 v10, v57, v10\n v_bcnt_u32_b32 v11, v65, v11\n v_xor_b32 v42, v26, v42\n v_xor_b32 v50, v26, v50\n v_xor_b32 v58, v26, v58\n v_xor_b32 v66, v26, v66\n v_bcnt_u32_b32 v8, v42, v8\n v_bcnt_u32_b32 v9, v50, v9\n v_bcnt_u32_b32 v10, v58, v10\n v_bcnt_u32_b32 v11, v66, v11\n v_xor_b32 v43, v27, v43\n v_xor_b32 v51, v27, v51\n v_xor_b32 v59, v27, v59\n v_xor_b32 v67, v27, v67\n v_bcnt_u32_b32 v8, v43, v8\n v_bcnt_u32_b32 v9, v51, v9\n v_bcnt_u32_b32 v10, v59, v10\n v_bcnt_u32_b32 v11, v67, v11\n v_xor_b32 v44, v28, v44\n v_xor_b32 v52, v28, v52\n v_xor_b32 v60, v28, v60\n v_xor_b32 v68, v28, v68\n v_bcnt_u32_b32 v8, v44, v8\n v_bcnt_u32_b32 v9, v52, v9\n v_bcnt_u32_b32 v10, v60, v10\n v_bcnt_u32_b32 v11, v68, v11\n v_xor_b32 v45, v29, v45\n v_xor_b32 v53, v29, v53\n v_xor_b32 v61, v29, v61\n v_xor_b32 v69, v29, v69\n v_bcnt_u32_b32 v8, v45, v8\n v_bcnt_u32_b32 v9, v53, v9\n v_bcnt_u32_b32 v10, v61, v10\n v_bcnt_u32_b32 v11, v69, v11\n v_xor_b32 v46, v30, v46\n v_xor_b32 v54, v30, v54\n v_xor_b32 v62, v30, v62\n v_xor_b32 v70, v30, v70\n v_bcnt_u32_b32 v8, v46, v8\n v_bcnt_u32_b32 v9, v54, v9\n v_bcnt_u32_b32 v10, v62, v10\n v_bcnt_u32_b32 v11, v70, v11\n v_xor_b32 v47, v31, v47\n v_xor_b32 v55, v31, v55\n v_xor_b32 v63, v31, v63\n v_xor_b32 v71, v31, v71\n v_bcnt_u32_b32 v8, v47, v8\n v_bcnt_u32_b32 v9, v55, v9\n v_bcnt_u32_b32 v10, v63, v10\n v_bcnt_u32_b32 v11, v71, v11\n"
#define K(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                           \
    {                                                                                                   \
        uint32_t r;                                                                                     \
        asm volatile("v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n"    \
                     "v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n v_mov_b32 v15, %1\n"  \
                     "v_mov_b32 v16, %1\n v_mov_b32 v17, %1\n v_mov_b32 v18, %1\n v_mov_b32 v19, %1\n"  \
                     "v_mov_b32 v20, %1\n v_mov_b32 v21, %1\n v_mov_b32 v22, %1\n v_mov_b32 v23, %1\n"  \
                     "s_movk_i32 s20, 0\n s_mov_b32 s21, 0x1234567\n s_mov_b32 s22, 0x89abcdef\n s_mov_b32 s23, 0x0f1e2d3c\n s_mov_b32 s24, 0x55aa1177\n v_mov_b32 v32, %1\n v_mov_b32 v33, %1\n v_mov_b32 v34, %1\n v_mov_b32 v35, %1\n v_mov_b32 v24, %1\n v_mov_b32 v25, %1\n v_mov_b32 v26, %1\n v_mov_b32 v27, %1\n v_mov_b32 v28, %1\n v_mov_b32 v29, %1\n v_mov_b32 v30, %1\n v_mov_b32 v31, %1\n"                                                              \
                     "1:\n" BODY BODY BODY BODY                                                         \
                     "s_add_u32 s20, s20, 1\n s_cmp_lt_u32 s20, %2\n s_cbranch_scc1 1b\n"               \
                     "v_xor_b32 %0, v8, v9\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v15\n"           \
                     : "=v"(r) : "v"(seed + threadIdx.x), "s"(ITERS / 4)                                \
                     : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
                       "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "s20", "s21", "s22", "s23", "s24", "scc");                                       \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                        \
    }
#define K1(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                           \
    {                                                                                                   \
        uint32_t r;                                                                                     \
        asm volatile("v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n"    \
                     "v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n v_mov_b32 v15, %1\n"  \
                     "v_mov_b32 v16, %1\n v_mov_b32 v17, %1\n v_mov_b32 v18, %1\n v_mov_b32 v19, %1\n"  \
                     "v_mov_b32 v20, %1\n v_mov_b32 v21, %1\n v_mov_b32 v22, %1\n v_mov_b32 v23, %1\n"  \
                     "s_movk_i32 s20, 0\n s_mov_b32 s21, 0x1234567\n s_mov_b32 s22, 0x89abcdef\n s_mov_b32 s23, 0x0f1e2d3c\n s_mov_b32 s24, 0x55aa1177\n v_mov_b32 v32, %1\n v_mov_b32 v33, %1\n v_mov_b32 v34, %1\n v_mov_b32 v35, %1\n v_mov_b32 v24, %1\n v_mov_b32 v25, %1\n v_mov_b32 v26, %1\n v_mov_b32 v27, %1\n v_mov_b32 v28, %1\n v_mov_b32 v29, %1\n v_mov_b32 v30, %1\n v_mov_b32 v31, %1\n"                                                              \
                     "1:\n" BODY                                                         \
                     "s_add_u32 s20, s20, 1\n s_cmp_lt_u32 s20, %2\n s_cbranch_scc1 1b\n"               \
                     "v_xor_b32 %0, v8, v9\n v_xor_b32 %0, %0, v10\n v_xor_b32 %0, %0, v15\n"           \
                     : "=v"(r) : "v"(seed + threadIdx.x), "s"(ITERS)                                \
                     : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
                       "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "s20", "s21", "s22", "s23", "s24", "scc");                                       \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                        \
    }
K(b_same, BODY_SAME) K(b_diff, BODY_DIFF) K(x_same, XBODY_SAME) K(x_diff, XBODY_DIFF)
K(h_sgpr, HBODY_SGPR) K(h_vgpr, HBODY_VGPR) K(h_inpl, HBODY_INPLACE) K(h_x, HBODY_XONLY) K(h_b, HBODY_BONLY)
K(h_ipv, HBODY_IPV) K(h_ipvb, HBODY_IPV_BATCH) K(h_ipvs, HBODY_IPV_SAMEACC) K(h_ipvx, HBODY_IPV_XONLY)
K(l_batch, LBODY) K(l_inter, LBODY2)
K1(h1_ipvs, HBODY_IPV_SAMEACC) K1(h1_ipv, HBODY_IPV) K1(h1_sgpr, HBODY_SGPR)
K(h_s0, HBODY_S0) K(h_s0one, HBODY_S0_ONE)
K(nb_x, NB_X) K(nb_xb, NB_XB) K(nb_b, NB_B) K(nb_sx, NB_SX) K(nb_sxb, NB_SXB)
K(m_same, MBODY_SAME) K(m_diff, MBODY_DIFF) K(p_same, PBODY_SAME) K(p_diff, PBODY_DIFF)
typedef void (*kern_t)(uint32_t *, uint32_t);
int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, w = argc > 1 ? atoi(argv[1]) : 8, blocks = cus * w;
    uint32_t *out;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    struct { const char *n; kern_t k; int per; } es[] = {
        {"bcnt srcs same bank", b_same, 8}, {"bcnt srcs different banks", b_diff, 8}, {"xor same bank", x_same, 8},
        {"xor different banks", x_diff, 8}, {"mul_f32 same bank", m_same, 8}, {"mul_f32 different banks", m_diff, 8},
        {"pk_mul same bank", p_same, 4}, {"8 x (t=s^row; acc=bcnt(t)+acc)", h_sgpr, 16}, {"8 x (t=v^row; acc=bcnt(t)+acc)", h_vgpr, 16}, {"8 x (row^=s; acc=bcnt(row)+acc)", h_inpl, 16}, {"8 x t=s^row only", h_x, 8}, {"in-place vgpr xor, s_nop, bcnt", nb_x, 16}, {"in-place vgpr xor, s_nop, bcnt, s_nop", nb_xb, 16}, {"in-place vgpr xor, bcnt, s_nop", nb_b, 16}, {"t=s^row, s_nop, bcnt", nb_sx, 16}, {"t=s^row, s_nop, bcnt, s_nop", nb_sxb, 16}, {"(row=row^v [dst=src0]; acc=bcnt(row)+acc)", h_s0, 16}, {"valu_seq form: (a=a^c; a=bcnt(c)+a)", h_s0one, 16}, {"[16/loop] (r^=v; r=bcnt(c)+r) one reg", h1_ipvs, 4}, {"[16/loop] (row^=v; acc=bcnt(row)+acc)", h1_ipv, 4}, {"[16/loop] (t=s^row; acc=bcnt(t)+acc)", h1_sgpr, 4}, {"32 row^=v then 32 bcnt (4 chains)", l_batch, 64}, {"8 x (4 row^=v, 4 bcnt)", l_inter, 64}, {"8 x (row^=v; acc=bcnt(row)+acc)", h_ipv, 16}, {"8 row^=v then 8 acc=bcnt(row)+acc", h_ipvb, 16}, {"8 x (r^=v; r=bcnt(c)+r) one reg", h_ipvs, 16}, {"8 x row^=v only", h_ipvx, 8}, {"8 x acc=bcnt(t)+acc only", h_b, 8}, {"pk_mul different banks", p_diff, 4}};
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (auto &e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 8;
        printf("%-28s %8.4f ms  %6.2f cycles/instr/SIMD @2.4GHz\n", e.n, ms, ms * 1e-3 * 2.4e9 / ((double)ITERS * e.per * w));
    }
    return 0;
}
