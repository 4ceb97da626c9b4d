/* abi_c99.c -- include/crsdr.h is a C header: compiled as strict C99 (gcc -std=c99 -pedantic) and linked against libcrsdr.so
 * by tests/test_abi_symbols.py.  Without a device every entry point reports CRSDR_ENODEV (-4): there is no CPU fallback. */
#include "crsdr.h"
#include <stdio.h>
int main(void)
{
    int n = -1;
    int rc = crsdr_device_count(&n);
    printf("abi %d, rc %d, devices %d, err '%s'\n", crsdr_abi_version(), rc, n, crsdr_last_error());
    crsdr_plan_desc d = {0};
    d.nrows = 4; d.blocksize = 1024; d.mode = CRSDR_MODE_DIGITAL;
    crsdr_plan *p = 0;
    rc = crsdr_plan_create(&p, &d);
    printf("plan_create rc %d (%s)\n", rc, rc ? crsdr_last_error() : "ok");
    if (!rc) crsdr_plan_destroy(p);
    return 0;
}
