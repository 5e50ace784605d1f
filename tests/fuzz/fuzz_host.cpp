// Fuzz driver for the HOST side of the C ABI (SURVEY 5 "race detection / sanitizers": address + undefined-behaviour sanitizers on the CPU
// build only). Built with -fsanitize=fuzzer,address,undefined against build/asan/libhypergreco.so (same sanitizers, no GPU code paths
// are reached: every entry below works on a host-only key). Targets, selected by HG_FUZZ_TARGET:
//   json     the bytes are a witness file: hg_witness_from_json and hg_witness_from_json_bn254 [REF bfv-gkr/src/poly.rs:12-44,
//            test.rs:21-33] must load it or fail with an error string - never crash, never read out of bounds
//   verify   the bytes are a Goldilocks proof for the n=1024 reference witness: hg_verify must accept or reject [REF
//            sk_encryption_circuit.rs:462-517]
//   verifybn the same for hg_verify_bn254 and the bn254 reference witness
// Seeds: the reference's own fixtures (tests/golden) and the CPU oracle's proofs of them (written by `make fuzz`).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include "../../include/hg.h"

static hg_params P;
static hg_pk* PK = nullptr;
static hg_witness *W = nullptr, *WB = nullptr;
static int target = 0;
static std::string tmp_path;

extern "C" int LLVMFuzzerInitialize(int*, char***) {
    const char* t = getenv("HG_FUZZ_TARGET");
    target = !t || !strcmp(t, "json") ? 0 : (!strcmp(t, "verify") ? 1 : 2);
    const char* root = getenv("HG_FUZZ_ROOT");
    const std::string gold = std::string(root ? root : ".") + "/tests/golden/";
    if (hg_params_builtin(1024, 1, &P) != 0) { fprintf(stderr, "params: %s\n", hg_last_error()); abort(); }
    if (hg_setup(nullptr, &P, &PK) != 0) { fprintf(stderr, "setup: %s\n", hg_last_error()); abort(); }
    if (hg_witness_from_json(&P, (gold + "sk_enc_1024_1x27_65537.json").c_str(), &W) != 0) { fprintf(stderr, "witness: %s\n", hg_last_error()); abort(); }
    if (hg_witness_from_json_bn254(&P, (gold + "bn254_sk_enc_1024_1x27_65537.json").c_str(), &WB) != 0) { fprintf(stderr, "bn254 witness: %s\n", hg_last_error()); abort(); }
    tmp_path = "/dev/shm/hg_fuzz_" + std::to_string((long)getpid()) + ".json";
    return 0;
}

extern "C" int LLVMFuzzerTestOneInput(const uint8_t* data, size_t size) {
    if (target == 0) {
        FILE* f = fopen(tmp_path.c_str(), "wb");
        if (!f) abort();
        fwrite(data, 1, size, f);
        fclose(f);
        hg_witness* w = nullptr;
        if (hg_witness_from_json(&P, tmp_path.c_str(), &w) == 0) hg_witness_free(w);
        else if (!hg_last_error() || !*hg_last_error()) abort();   // a failure without a reason is a bug too
        w = nullptr;
        if (hg_witness_from_json_bn254(&P, tmp_path.c_str(), &w) == 0) hg_witness_free(w);
        else if (!hg_last_error() || !*hg_last_error()) abort();
    } else if (target == 1) {
        const int rc = hg_verify(PK, W, data, size);
        if (rc != 0 && (!hg_last_error() || !*hg_last_error())) abort();
    } else {
        const int rc = hg_verify_bn254(PK, WB, data, size);
        if (rc != 0 && (!hg_last_error() || !*hg_last_error())) abort();
    }
    return 0;
}
