/* ref_launch.c -- opens oracle/_ref/libref_driver.so lazily (its four unresolved reference
 * symbols are never called) and runs one job file.  TEST INFRASTRUCTURE ONLY. */
#include <dlfcn.h>
#include <libgen.h>
#include <stdio.h>
#include <string.h>
int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: ref_driver <jobfile>\n"); return 2; }
  char path[4096], self[4096];
  strncpy(self, argv[0], sizeof(self) - 1);
  self[sizeof(self) - 1] = 0;
  snprintf(path, sizeof(path), "%s/libref_driver.so", dirname(self));
  void *h = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
  int (*fn)(const char *) = (int (*)(const char *))dlsym(h, "ref_driver_main");
  if (!fn) { fprintf(stderr, "%s\n", dlerror()); return 2; }
  return fn(argv[1]);
}
