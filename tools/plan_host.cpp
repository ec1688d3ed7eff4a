// A host with no Python in it: load a compiled forward plan + the packed weights + one frame of inputs, run the
// forward through the C ABI (tt_plan_load / tt_plan_bind / tt_encoder_fwd / tt_decoder_fwd) and write the result tensors.
//
//   plan_host <dir> [repeats]
//     <dir>/plan.bin, <dir>/weights.bin, <dir>/input<i>.bin   (written by thinktwice_amd.plan.ForwardPlan.save)
//     -> <dir>/out_<name>.bin (dense f32, the output's logical shape) and a line per output + the ms per forward on stdout.
// Build: hipcc -O2 -I include tools/plan_host.cpp -L thinktwice_amd -lthinktwice_hip -Wl,-rpath,$PWD/thinktwice_amd -o tools/plan_host
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <string>
#include <vector>

#include "thinktwice_hip.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 2;                                                              \
        }                                                                          \
    } while (0)
#define TT_OK(x)                                                                   \
    do {                                                                           \
        if ((x) != 0) {                                                            \
            fprintf(stderr, "%s failed: %s\n", #x, tt_last_error());               \
            return 3;                                                              \
        }                                                                          \
    } while (0)

static bool read_file(const std::string& path, std::vector<char>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    const bool ok = fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: plan_host <dir> [repeats]\n");
        return 1;
    }
    const std::string dir = argv[1];
    const int repeats = argc > 2 ? atoi(argv[2]) : 5;
    tt_plan* plan = tt_plan_load((dir + "/plan.bin").c_str());
    if (!plan) {
        fprintf(stderr, "tt_plan_load: %s\n", tt_last_error());
        return 3;
    }
    const int nb = tt_plan_num_buffers(plan), ns = tt_plan_num_streams(plan);
    printf("plan: %d calls, %d buffers, %d streams\n", tt_plan_num_calls(plan), nb, ns);
    std::vector<void*> bases((size_t)nb, nullptr);
    for (int i = 0; i < nb; ++i) {
        const long long bytes = tt_plan_buffer_bytes(plan, i);
        HIP_OK(hipMalloc(&bases[(size_t)i], (size_t)(bytes > 0 ? bytes : 256)));
        std::vector<char> host;
        const std::string src = i == 0 ? dir + "/weights.bin" : (i >= 2 ? dir + "/input" + std::to_string(i - 2) + ".bin" : "");
        if (!src.empty()) {
            if (!read_file(src, host) || (long long)host.size() > bytes) {
                fprintf(stderr, "cannot read %s (buffer %d '%s', %lld bytes)\n", src.c_str(), i, tt_plan_buffer_name(plan, i), bytes);
                return 4;
            }
            HIP_OK(hipMemcpy(bases[(size_t)i], host.data(), host.size(), hipMemcpyHostToDevice));
        }
    }
    TT_OK(tt_plan_bind(plan, bases.data(), nb));
    std::vector<void*> streams((size_t)ns, nullptr);
    for (int i = 0; i < ns; ++i) {
        hipStream_t s;
        HIP_OK(hipStreamCreate(&s));
        streams[(size_t)i] = s;
    }
    double ms = 0.0;
    for (int r = 0; r < repeats + 1; ++r) {            // (first pass: warm-up)
        HIP_OK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        TT_OK(tt_encoder_fwd(plan, streams.data(), ns));
        TT_OK(tt_decoder_fwd(plan, streams.data(), ns));
        HIP_OK(hipDeviceSynchronize());
        if (tt_device_faults() != 0) {                 // a wide-chain barrier gave up: the outputs are NaN, say so
            fprintf(stderr, "forward %d: a tt_mlp_chain_wide barrier timed out (tt_device_faults) -- outputs invalid\n", r);
            return 6;
        }
        if (r > 0) ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("forward: %.3f ms (mean of %d, host wall clock)\n", ms / (repeats > 0 ? repeats : 1), repeats);
    for (int i = 0; i < tt_plan_num_outputs(plan); ++i) {
        const char* name;
        int buffer, ndim;
        long long offset, shape[8], stride[8];
        TT_OK(tt_plan_output(plan, i, &name, &buffer, &offset, &ndim, shape, stride));
        if (buffer < 0) continue;                      // (markers)
        long long n = 1, span = 1;
        for (int d = 0; d < ndim; ++d) {
            n *= shape[d];
            span += (shape[d] - 1) * stride[d];
        }
        std::vector<float> raw((size_t)span), dense((size_t)n);
        HIP_OK(hipMemcpy(raw.data(), (char*)bases[(size_t)buffer] + offset, (size_t)span * 4, hipMemcpyDeviceToHost));
        for (long long k = 0; k < n; ++k) {            // gather the (possibly strided) view into its logical order
            long long rem = k, src = 0;
            for (int d = ndim - 1; d >= 0; --d) {
                src += (rem % shape[d]) * stride[d];
                rem /= shape[d];
            }
            dense[(size_t)k] = raw[(size_t)src];
        }
        FILE* f = fopen((dir + "/out_" + name + ".bin").c_str(), "wb");
        if (!f || fwrite(dense.data(), 4, (size_t)n, f) != (size_t)n) return 5;
        fclose(f);
        printf("output %s: %lld floats\n", name, n);
    }
    tt_plan_destroy(plan);
    return 0;
}
