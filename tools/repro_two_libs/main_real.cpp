// The same order experiment with the REAL libraries and no Python: dlopen the stock libfibhip.so and a specialised build
// (fib_tf_amd/_spec/libfibhip_br_*.so), call fibhip_warm (one empty kernel of that build) in the order given: "s" = stock,
// "b" = specialised.  "bs" is the order that dies under rocprofv3 from Python; "sb" is what the binding enforces.
#include <dlfcn.h>
#include <cstdio>
int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: main_real <stock.so> <spec.so> <order>\n"); return 2; }
    void *s = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL), *b = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!s || !b) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    int (*ws)(int) = (int (*)(int))dlsym(s, "fibhip_warm"), (*wb)(int) = (int (*)(int))dlsym(b, "fibhip_warm");
    for (const char *c = argv[3]; *c; ++c) {
        const int rc = *c == 's' ? ws(0) : wb(0);
        printf("fibhip_warm of the %s build -> %d\n", *c == 's' ? "stock" : "specialised", rc);
        fflush(stdout);
    }
    return 0;
}
