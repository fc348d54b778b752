/* Test-side tool (never linked into the product): LD_PRELOAD counter of the allocations of sizeof(Gene) = 384 bytes under the
 * reference binary oracle/_ref/augustus_ref -- the evidence for the order of alternatives with EQUAL mean state probability
 * (the reference sorts Transcript POINTERS, src/gene.cc:3196).
 *   gcc -O2 -shared -fPIC -o /tmp/gene_alloc_trace.so tests/golden/gene_alloc_trace.c -ldl
 *   LD_PRELOAD=/tmp/gene_alloc_trace.so oracle/_ref/augustus_ref --species=human --UTR=on --sample=30 ... x.fa 2> trace.txt
 * One line per allocation: "M <malloc count> <address>".  Within the sampling loop of the first record of a run (the lines a few
 * ten thousand mallocs apart) the addresses fall. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void *(*real_malloc)(size_t);
static unsigned long nmalloc;
void *malloc(size_t n) {
    if (!real_malloc) real_malloc = (void *(*)(size_t))dlsym(RTLD_NEXT, "malloc");
    void *p = real_malloc(n);
    nmalloc++;
    if (n == 384) {
        char b[96];
        int k = snprintf(b, sizeof b, "M %lu %p\n", nmalloc, p);
        if (write(2, b, (size_t)k) < 0) return p;
    }
    return p;
}
