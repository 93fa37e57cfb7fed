/* TEST HARNESS (not product): a plain C99 consumer of include/bzk.h - what a cgo / Rust-FFI / JNI binding is underneath.  Links against
 * libbzk.so and uses only host-side entry points (no GPU): the status strings, SHA3, the bincode work decoder and the Groth16 verifier.
 *   abi_consumer vk.bin inputs.bin proof.bin work.bin prover.bin
 * prints one line per check; exit code 0 iff every check holds.  Driven by tests/test_abi_cpu.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bzk.h"

static unsigned char* slurp(const char* path, uint64_t* len) {
    FILE* f = fopen(path, "rb");
    unsigned char* buf;
    long n;
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf = (unsigned char*)malloc((size_t)n + 1);
    if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = NULL; }
    fclose(f);
    *len = (uint64_t)n;
    return buf;
}

int main(int argc, char** argv) {
    static const unsigned char sha3_empty[32] = {0xa7, 0xff, 0xc6, 0xf8, 0xbf, 0x1e, 0xd7, 0x66, 0x51, 0xc1, 0x47, 0x56, 0xa0, 0x61, 0xd6, 0x62,
                                                 0xf5, 0x80, 0xff, 0x4d, 0xe4, 0x3b, 0x49, 0xfa, 0x82, 0xd8, 0x0a, 0x4b, 0x80, 0xf8, 0x43, 0x4a};
    unsigned char digest[32], enc[391];
    uint64_t vk_len, in_len, pr_len, wk_len, pv_len, used = 0, info[12];
    unsigned char *vk, *in, *pr, *wk, *pv;
    bzk_mpn_work* work = NULL;
    int bad = 0;
    int32_t st;
    if (argc != 6) return 2;
    printf("abi version %u, status 0 = \"%s\", status %d = \"%s\"\n", bzk_abi_version(), bzk_strerror(BZK_OK), BZK_E_ARG, bzk_strerror(BZK_E_ARG));
    st = bzk_host_sha3_256(NULL, 0, digest);
    printf("sha3_256(\"\") %s\n", st == BZK_OK && !memcmp(digest, sha3_empty, 32) ? "ok" : "WRONG");
    bad |= !(st == BZK_OK && !memcmp(digest, sha3_empty, 32));
    vk = slurp(argv[1], &vk_len); in = slurp(argv[2], &in_len); pr = slurp(argv[3], &pr_len); wk = slurp(argv[4], &wk_len); pv = slurp(argv[5], &pv_len);
    if (!vk || !in || !pr || !wk || !pv || pr_len != 387 || pv_len != 32 || in_len % 32) return 2;
    st = bzk_groth16_verify(vk, vk_len, in, (uint32_t)(in_len / 32), pr);
    printf("groth16_verify(valid) = %d\n", (int)st);
    bad |= st != 1;
    in[0] ^= 1;  /* another public input */
    st = bzk_groth16_verify(vk, vk_len, in, (uint32_t)(in_len / 32), pr);
    printf("groth16_verify(wrong input) = %d\n", (int)st);
    bad |= st != 0;
    st = bzk_groth16_verify(vk, vk_len - 1, in, (uint32_t)(in_len / 32), pr);
    printf("groth16_verify(truncated key) = %d\n", (int)st);
    bad |= st != 0;
    st = bzk_groth16_verify(NULL, 0, in, 0, pr);
    printf("groth16_verify(null key) = %d (%s)\n", (int)st, bzk_strerror(st));
    bad |= st != BZK_E_ARG;
    st = bzk_zkproof_encode(pr, enc);
    printf("zkproof_encode: %d, %d bytes, variant tag %u\n", (int)st, (int)sizeof enc, (unsigned)enc[0]);
    bad |= st != BZK_OK;
    st = bzk_mpn_work_decode(wk, wk_len, 0, &work, &used);
    printf("mpn_work_decode: %d, consumed %llu of %llu\n", (int)st, (unsigned long long)used, (unsigned long long)wk_len);
    bad |= !(st == BZK_OK && work && used == wk_len);
    if (work) {
        st = bzk_mpn_work_info(work, info);
        printf("mpn_work_info: %d, kind %llu, log4 tree %llu, log4 token tree %llu, log4 batch %llu\n", (int)st, (unsigned long long)info[0],
               (unsigned long long)info[1], (unsigned long long)info[2], (unsigned long long)info[3]);
        bad |= st != BZK_OK;
        st = bzk_mpn_work_verify(work, pv, pr);  /* a proof of another statement: must be refused, not crash */
        printf("mpn_work_verify(foreign proof) = %d\n", (int)st);
        bad |= st != 0;
        bzk_mpn_work_free(work);
    }
    st = bzk_mpn_work_decode(wk, wk_len / 2, 0, &work, &used);
    printf("mpn_work_decode(truncated) = %d (%s)\n", (int)st, bzk_strerror(st));
    bad |= st == BZK_OK;
    free(vk); free(in); free(pr); free(wk); free(pv);
    printf(bad ? "FAILED\n" : "all checks hold\n");
    return bad;
}
