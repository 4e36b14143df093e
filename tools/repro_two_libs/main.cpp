// dlopen liba.so, dlopen libb.so, then launch in the order given on the command line (e.g. "ba" = b's kernel first, then a's).
// tools/repro_two_libs/run.sh runs "ba" and "ab" once each under `rocprofv3 --kernel-trace`.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
int main(int argc, char **argv)
{
    const char *order = argc > 1 ? argv[1] : "ba";
    void *a = dlopen("./liba.so", RTLD_NOW | RTLD_LOCAL), *b = dlopen("./libb.so", RTLD_NOW | RTLD_LOCAL);
    if (!a || !b) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    int (*la)(void) = (int (*)(void))dlsym(a, "launch_a"), (*lb)(void) = (int (*)(void))dlsym(b, "launch_b");
    for (const char *c = order; *c; ++c) {
        const int rc = *c == 'a' ? la() : lb();
        printf("launch_%c -> %d\n", *c, rc);
        fflush(stdout);
    }
    return 0;
}
