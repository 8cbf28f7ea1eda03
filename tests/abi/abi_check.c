/* The drop-in boundary from plain C: include/nerf_hip.h must compile as C (not only as C++), and libnerf_hip.so must link
 * and answer the host-only entry points (sizes, option errors) without a GPU.  Built and run by tests/test_host_cpu.py. */
#include <stdio.h>
#include <string.h>
#include "nerf_hip.h"

static int check(int ok, const char* what) {
  if (!ok) fprintf(stderr, "abi_check: FAILED: %s\n", what);
  return ok ? 0 : 1;
}

int main(void) {
  int bad = 0;
  nerf_mlp_arch a;
  memset(&a, 0, sizeof a);
  a.n_layers = 8; a.width = 256; a.in_pos = 63; a.in_dir = 27; a.skip_layer = 4; a.use_viewdirs = 1; a.out_ch = 4;
  bad += check(nerf_abi_version() == NERF_ABI_VERSION, "nerf_abi_version() == NERF_ABI_VERSION");
  a.precision = 16;
  const long long p16 = (long long)nerf_mlp_packed_bytes(&a);
  bad += check(nerf_mlp_param_count(&a) == 595844, "8 x 256 view model has 595 844 parameters");
  a.precision = 32;
  const long long p32 = (long long)nerf_mlp_packed_bytes(&a);
  a.precision = 22;
  const long long p22 = (long long)nerf_mlp_packed_bytes(&a);
  bad += check(p16 > 0 && p32 > p16 && p22 > p16, "packed image sizes per precision");
  bad += check(nerf_mlp_acts_bytes(&a, 65) == 8ll * 325 * 1024, "precision-22 activation workspace: 8 padded tiles x 325 KiB");
  a.precision = 7;
  bad += check(nerf_mlp_packed_bytes(&a) < 0, "unknown precision is refused");
  bad += check(nerf_set_option("no_such_option", 1) == NERF_E_UNSUPPORTED, "unknown option -> NERF_E_UNSUPPORTED");
  bad += check(strstr(nerf_last_error(), "no_such_option") != NULL, "nerf_last_error names the key");
  bad += check(nerf_sh_encode(NULL, 4, 2, NULL, NULL) == NERF_E_NULL, "NULL pointer -> NERF_E_NULL");
  if (!bad) printf("abi_check ok: packed bytes %lld (bf16) %lld (fp32) %lld (precision 22)\n", p16, p32, p22);
  return bad ? 1 : 0;
}
