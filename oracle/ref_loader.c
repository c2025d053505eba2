/* ref_loader.c -- opens oracle/_ref/libref_driver.so with lazy binding for Python callers.
 * TEST INFRASTRUCTURE ONLY.  ctypes always adds RTLD_NOW, which would demand the four reference symbols that live in
 * files needing cuda_runtime.h (see ref_driver.cc); they are never reached, so the driver is opened RTLD_LAZY here and
 * Python calls it through the function pointers this returns. */
#include <dlfcn.h>
#include <stdio.h>

void *ref_loader_open(const char *path) {
  void *h = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
  if (!h) fprintf(stderr, "ref_loader: %s\n", dlerror());
  return h;
}

void *ref_loader_sym(void *handle, const char *name) { return handle ? dlsym(handle, name) : 0; }
