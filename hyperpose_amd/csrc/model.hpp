// model.hpp - the hp_model object behind hp_model_* (include/hp_hip.h): a layer list over numbered tensors, the output
// descriptors and the input normalisation.  Filled by the built-in topology builders (models.cpp) or by the ONNX
// importer (onnx_import.cpp).
#pragma once
#include "hp_common.hpp"

#include <cstring>
#include <string>
#include <vector>

// output size of a layer on an H x W input (TF "SAME" or the explicit ONNX-style pads; same rule as engine.cpp pass 1)
inline void hp_layer_out_size(const hp_layer& L, int H, int W, int& OH, int& OW)
{
    if (L.op == HP_OP_UPSAMPLE) {
        OH = H * L.stride, OW = W * L.stride;
    } else if (L.pad_explicit) {
        OH = (H + L.pad[0] + L.pad[2] - ((L.kh - 1) * L.dil + 1)) / L.stride + 1;
        OW = (W + L.pad[1] + L.pad[3] - ((L.kw - 1) * L.dil + 1)) / L.stride + 1;
    } else {
        OH = (H + L.stride - 1) / L.stride, OW = (W + L.stride - 1) / L.stride;
    }
}

struct hp_model {
    std::string arch;
    int in_w = 0, in_h = 0;
    std::vector<hp_layer> layers;
    std::vector<float> init_scale; // per layer multiplier on the He std (final linear heads are kept small)
    std::vector<float> init_bias;  // per layer constant added to every bias (keeps sigmoid heads sparse with synthetic weights)
    std::vector<hp_output_desc> outputs;
    int64_t n_weights = 0;
    int next_tensor = 1;
    float mean[3] = { 0, 0, 0 }, inv_std[3] = { 1, 1, 1 };
    std::vector<float> weights; // imported models (onnx_import.cpp) carry their weights; built-in topologies do not

    int new_tensor() { return next_tensor++; }

    // generic layer append; returns the output tensor id
    int add(int op, int in, int in_coff, int cin, int cout, int k, int stride, int dil, int act, bool bias,
        int out = -1, int out_coff = 0, int res = -1, int res_before_act = 0, float scale = 1.f, float act_param = 0.f)
    {
        hp_layer L;
        memset(&L, 0, sizeof(L));
        L.op = op, L.in = in, L.in_coff = in_coff, L.res = res, L.res_before_act = res_before_act;
        L.out = out < 0 ? new_tensor() : out, L.out_coff = out_coff;
        L.cin = cin, L.cout = cout, L.kh = k, L.kw = k, L.stride = stride, L.dil = dil, L.act = act, L.act_param = act_param;
        L.w_off = -1, L.b_off = -1, L.alpha_off = -1;
        if (op == HP_OP_CONV) {
            L.w_off = n_weights;
            n_weights += (int64_t)cout * k * k * cin;
        } else if (op == HP_OP_DWCONV) {
            L.w_off = n_weights;
            n_weights += (int64_t)cin * k * k;
        }
        if (bias && op != HP_OP_MAXPOOL && op != HP_OP_UPSAMPLE) {
            L.b_off = n_weights;
            n_weights += cout;
        }
        if (act == HP_ACT_PRELU) {
            L.alpha_off = n_weights;
            n_weights += cout;
        }
        layers.push_back(L);
        init_scale.push_back(scale);
        init_bias.push_back(0.f);
        return L.out;
    }
    int conv(int in, int cin, int cout, int k, int act, int stride = 1, int dil = 1, int in_coff = 0)
    {
        return add(HP_OP_CONV, in, in_coff, cin, cout, k, stride, dil, act, true);
    }
    int dw_block(int in, int cin, int cout, int stride = 1, int dil = 1)
    {
        // dw_conv_block: DepthwiseConv2d(b=None)+BN+ReLU, Conv2d 1x1 (b=None)+BN+ReLU  (backbones.py:190-197)
        const int t = add(HP_OP_DWCONV, in, 0, cin, cin, 3, stride, dil, HP_ACT_RELU, true);
        return add(HP_OP_CONV, t, 0, cin, cout, 1, 1, 1, HP_ACT_RELU, true);
    }
    int pool(int in, int c, int k, int stride) { return add(HP_OP_MAXPOOL, in, 0, c, c, k, stride, 1, HP_ACT_NONE, false); }
    void output(const char* name, int tensor, int coff, int channels, int act = HP_ACT_NONE)
    {
        hp_output_desc o;
        memset(&o, 0, sizeof(o));
        strncpy(o.name, name, sizeof(o.name) - 1);
        o.tensor = tensor, o.coff = coff, o.channels = channels, o.act = act;
        outputs.push_back(o);
    }
};
