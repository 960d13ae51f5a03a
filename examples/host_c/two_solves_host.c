/* Two reconstructions AT THE SAME TIME from one plain C process: two host threads, each with its own HIP stream, device
 * buffers, side stream + events and range-guard word (pnp_solve.h), driving the iteration-level C ABI concurrently.
 * The second solve's measurement is scaled by 2e4 so that its activations leave fp16's range: ITS word must come back
 * set, the first solve's word and the library's process-wide word must stay clear, and the first solve's result must be
 * bit-identical to the same solve run alone (tests/test_gpu_cabi_host.py compares the files).
 *
 *   two_solves_host <problem.bin> <out_a.bin> <out_b.bin>
 */
#include <pthread.h>

#include "pnp_solve.h"

static void* worker(void* p) { pnp_solve((pnp_solve_t*)p); return NULL; }

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s problem.bin out_a.bin out_b.bin\n", argv[0]); return 1; }
    pnp_solve_t jobs[2];
    memset(jobs, 0, sizeof jobs);
    for (int i = 0; i < 2; ++i) {
        jobs[i].problem = argv[1]; jobs[i].out = argv[2 + i];
        jobs[i].two_streams = 1;
        jobs[i].input_scale = i == 0 ? 1.0f : 2e4f;
    }
    pthread_t th[2];
    for (int i = 0; i < 2; ++i)
        if (pthread_create(&th[i], NULL, worker, &jobs[i]) != 0) { perror("pthread_create"); return 1; }
    for (int i = 0; i < 2; ++i) pthread_join(th[i], NULL);
    int global_word = 0;
    SCICHK(scipnp_split_overflow(0, &global_word, NULL));          /* this thread bound nothing: the process-wide word */
    printf("solve A overflow %d, solve B overflow %d, process-wide word %d\n", jobs[0].overflow, jobs[1].overflow, global_word);
    return (jobs[0].overflow == 0 && jobs[1].overflow == 1 && global_word == 0) ? 0 : 6;
}
