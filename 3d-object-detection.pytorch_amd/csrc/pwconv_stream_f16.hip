// fp16-storage instantiation of the streaming pointwise kernel: INFERENCE forward only (pwconv_stream.hip has the code; this
// unit selects the storage type, v_mfma_f32_16x16x32_f16 and the entry name t3d_pw::stream_launch_f16).
#define T3D_PW_F16 1
#include "pwconv_stream.hip"
