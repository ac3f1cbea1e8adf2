// tools/direct_timeline.hip — block timeline (s_memtime) and long-run rate of conv_direct_kernel on one OpenPose-VGG19 stage layer
// (7x7 128 -> 128 at 16 x 54 x 96), run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/direct_timeline.hip -Lhyperpose_amd -lhp_hip -Wl,-rpath,$PWD/hyperpose_amd -o gpurun_out/direct_timeline
#include "../hyperpose_amd/csrc/conv_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int KS = argc > 1 ? atoi(argv[1]) : 7, CIN = argc > 2 ? atoi(argv[2]) : 128, COUT = argc > 3 ? atoi(argv[3]) : 128;
    const int B = argc > 4 ? atoi(argv[4]) : 16, H = argc > 5 ? atoi(argv[5]) : 54, W = argc > 6 ? atoi(argv[6]) : 96;
    const int P = KS / 2, cs = CIN, wp = W + 2 * P, img = (H + 2 * P) * wp;
    const size_t in_elems = (size_t)B * img * cs, out_elems = (size_t)B * H * W * COUT;
    std::vector<__half> hin(in_elems), hw((size_t)KS * KS * COUT * CIN);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hin) v = __float2half(rnd());
    for (auto& v : hw) v = __float2half(rnd() * 0.05f);
    __half *din, *dw, *dout;
    float* dbias;
    unsigned long long* dbg;
    CK(hipMalloc(&din, in_elems * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dout, out_elems * 2));
    CK(hipMalloc(&dbias, COUT * 4)); CK(hipMalloc(&dbg, 64 * 8));
    CK(hipMemcpy(din, hin.data(), in_elems * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dbias, 0, COUT * 4)); CK(hipMemset(dbg, 0, 64 * 8));
    hp::conv_params p{};
    p.in = hp::tview{ din + ((size_t)P * wp + P) * cs, cs, 0, wp, img };
    p.B = B, p.H = H, p.W = W, p.OH = H, p.OW = W, p.Cin = CIN, p.Cout = COUT, p.Cout_pad = COUT;
    p.KH = p.KW = KS, p.stride = 1, p.dil = 1, p.pad_t = p.pad_l = P;
    p.w = dw, p.bias = dbias, p.alpha = nullptr, p.act = hp::ACT_RELU, p.act_param = 0;
    p.res = hp::tview{ nullptr, 0, 0, 0, 0 };
    p.out = hp::tview{ dout, COUT, 0, W, H * W };
    p.out_f32 = nullptr, p.npix = B * H * W, p.dbg = nullptr;
    hp::set_act(p);
    p.w_layout = hp::conv_weight_layout(p);
    printf("w_layout %d tile %d\n", p.w_layout, hp::conv_mfma_tile(p));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int i = 0; i < 20; ++i) CK(hp::launch_conv_mfma(p, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) CK(hp::launch_conv_mfma(p, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * B * H * W * COUT * KS * KS * CIN;
    printf("%dx%d %d->%d  %d x %d x %d: %.1f us per launch, %.1f TFLOP/s\n", KS, KS, CIN, COUT, B, H, W, ms / iters * 1e3, fl / (ms / iters * 1e-3) / 1e12);
    p.dbg = dbg;
    CK(hp::launch_conv_mfma(p, st));
    CK(hipStreamSynchronize(st));
    unsigned long long h[64];
    CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    for (int w = 0; w < 2; ++w) {
        printf("wave %d stamps (cycles since block start): ", w ? 7 : 0);
        for (int i = 0; i < 32 && h[w * 32 + i]; ++i) printf("%llu ", h[w * 32 + i] - h[0]);
        printf("\n");
    }
    return 0;
}
