// Shared by qbnn_f32.hip (the gather forms: any geometry) and qbnn_q8t.hip (the LDS-tiled form of the ResNet's 3 x 3 convs): the argument block
// of the QAT convs on the int8 matrix pipe (reference quantized/conv_qat.py:139-167 in eval: both operands are fake-quantised tensors, i.e.
// integers on a per-sample grid).
#ifndef QBNN_Q8_H_
#define QBNN_Q8_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ConvQ8Args {
  const int8_t* x; int64_t x_ss;     // centred activations m_x [S][B][H][W][Cin]
  const int8_t* w; int64_t w_ss;     // raw weights q_w [S][Cout][KH][KW][Cin]
  const float* s_x; const float* s_w; const int* z_w;      // per sample
  float* y; int64_t y_ss;
  int B, H, W, Cin, Cout, KS, stride, pad, Ho, Wo, relu;
  const float* div; const float* bias; const float* alpha; const float* beta;
  float* mm_partials;
};
typedef int v16i_q8 __attribute__((ext_vector_type(16)));
typedef int v4i_q8 __attribute__((ext_vector_type(4)));
typedef int v2i_q8 __attribute__((ext_vector_type(2)));

// qbnn_q8t.hip: workgroups per sample of the tiled form for this geometry, 0 where it has none (the caller then takes a gather form)
int qbnn_conv_q8t_blocks(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad);
bool qbnn_conv_q8t_aligned(const ConvQ8Args& a);      // operand / output alignment that form needs
// launches it (grid = qbnn_conv_q8t_blocks x n_samples); the caller has checked pointers and qbnn_conv_q8t_aligned
int qbnn_launch_conv_q8t(const ConvQ8Args& a, int n_samples, hipStream_t st);
#endif
