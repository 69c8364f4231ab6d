// Times the PRODUCTION fused GRU step (csrc/gru.hip, included verbatim) in the same harness as exp_gru.hip.
#include "../inpaintnet_amd/csrc/gru.hip"
#include "../inpaintnet_amd/csrc/prof.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 512;
    float *hp, *W, *gi, *bh, *h2, *sv;
    (void)hipMalloc(&hp, (size_t)B * H * 4); (void)hipMalloc(&W, (size_t)3 * H * H * 4); (void)hipMalloc(&gi, (size_t)B * 3 * H * 4);
    (void)hipMalloc(&bh, 3 * H * 4); (void)hipMalloc(&h2, (size_t)B * H * 4); (void)hipMalloc(&sv, (size_t)5 * B * H * 4);
    std::vector<float> tmp((size_t)3 * H * H > (size_t)B * 3 * H ? (size_t)3 * H * H : (size_t)B * 3 * H);
    for (auto& x : tmp) x = (rand() % 2001 - 1000) * 1e-4f;
    (void)hipMemcpy(hp, tmp.data(), (size_t)B * H * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, tmp.data(), (size_t)3 * H * H * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(gi, tmp.data(), (size_t)B * 3 * H * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bh, tmp.data(), 3 * H * 4, hipMemcpyHostToDevice);
    for (int save = 0; save < 2; ++save) {
        GruFwdBatch bt{};
        bt.H = H; bt.nprob = 1;
        GruFwdProb& P = bt.p[0];
        P.B = B; P.h_prev = hp; P.ld_hprev = H; P.W_hh = W; P.b_hh = bh; P.gi_dense = gi; P.ld_gi = 3L * H;
        P.h_new = h2; P.ld_hnew = H;
        if (save) { long as = (long)B * H; P.sv_r = sv; P.sv_z = sv + as; P.sv_n = sv + 2 * as; P.sv_ghn = sv + 3 * as; P.sv_hprev = sv + 4 * as; }
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        for (int i = 0; i < 10; ++i) launch_gru_fwd(bt, 0);
        (void)hipEventRecord(a, 0);
        const int it = 200;
        for (int i = 0; i < it; ++i) launch_gru_fwd(bt, 0);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("production gru_step_fwd B=%d H=%d save=%d: %7.2f us per launch\n", B, H, save, ms * 1e3f / it);
    }
    return 0;
}
