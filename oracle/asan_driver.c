/* AddressSanitizer/UBSan run of the oracle on the CPU (test infrastructure; GPU ASan is not available
 * on the pool).  Folds a few sequences and evaluates one structure; prints a checksum. */
#include <stdio.h>
#include <string.h>
typedef struct OracleResult OracleResult;
OracleResult *oracle_fold(const char *, int, int, int, int, double, double, double, double, int *);
int oracle_result_n_steps(const OracleResult *);
int oracle_result_step_size(const OracleResult *, int);
int oracle_result_dcal(const OracleResult *, int, int);
const char *oracle_result_struct(const OracleResult *, int, int);
void oracle_result_free(OracleResult *);
int oracle_eval_structure(const char *, const char *, int *);

int main(void)
{
    const char *seqs[] = {
        "GGGUUUGCGGUGUAAGUGCAGCCCGUCUUACACCGUGCGGCACAGGCACUAGUACUGAUGUCGUAUACAGGGCUUUUGACAU",
        "GGGGAAUUAGCUCAAAUGGUAGAGCGCUCGCUUAGCAUGCGAGAGGUAGCGGGAUCGAUGCCCGCAUUCUCCACCA",
        "A", "GC", "GGGNNNNCCC", "ACGUACGUACGUACGUACGUACGUACGUACGUACGUACGUACGUACGUACGUACGUACGU"};
    long sum = 0;
    for (unsigned i = 0; i < sizeof seqs / sizeof *seqs; i++) {
        int err = 0;
        OracleResult *r = oracle_fold(seqs[i], 100, 20, 1000, 3, 0.0, 3.0, 2.0, 1.0, &err);
        if (!r) { printf("fold error %d\n", err); return 1; }
        for (int s = 0; s < oracle_result_n_steps(r); s++)
            for (int k = 0; k < oracle_result_step_size(r, s); k++) {
                int d = 0;
                if (oracle_eval_structure(seqs[i], oracle_result_struct(r, s, k), &d) || d != oracle_result_dcal(r, s, k)) {
                    printf("energy mismatch\n");
                    return 2;
                }
                sum += d;
            }
        oracle_result_free(r);
    }
    printf("asan ok %ld\n", sum);
    return 0;
}
