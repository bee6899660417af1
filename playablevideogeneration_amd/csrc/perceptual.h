// VGG19 perceptual loss (training/losses.py:379-491, model/layers/vgg.py:8-56): state kept in the context and the driver entry points.
#pragma once
#include "common.h"
#include "pack.h"

#define VGG_NCONV 13      // conv1_1 .. conv5_1: what model/layers/vgg.py:25-34 evaluates of torchvision's vgg19().features

struct VggLayer { PackDesc pd; float* wp; float* wpd; float* bias; int kd, cd_pad;
                  void* wq[3]; void* wqd[2]; };      // split 16-bit forms (conv_hx.hip): [0] two planes (3 products), [1] one plane; forward f16, dgrad bf16; wq[2]: forward as split bf16
struct VggState { bool enabled = false, loaded = false; VggLayer conv[VGG_NCONV]; };
struct VggLevels { double numel[3][5]; };      // elements of the level-l feature map at resolution r (N * C * H * W)

struct caddy_ctx;
struct caddy_param_info;
struct T4;
int vgg_param_count();
long vgg_param_floats();
int vgg_param_info(int index, caddy_param_info* out);
void vgg_build(caddy_ctx* c);
int vgg_load(caddy_ctx* c, const float* flat);
void vgg_perceptual(caddy_ctx* c, double lambda, const T4* gt_img, VggLevels* lv);
void vgg_gt_prefetch(caddy_ctx* c, int Trec, int t_off);
int vgg_eval_per_frame(caddy_ctx* c, double* out_host);      // evaluation: per reconstructed frame and level, full resolution (5 x N doubles, host)
