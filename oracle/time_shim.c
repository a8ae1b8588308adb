/* time_shim.c -- TEST INFRASTRUCTURE.  LD_PRELOAD this so that the unmodified reference
 * sampler's srand(time(0)) (gen_merw.cpp:161) becomes srand($PN_FAKE_TIME). */
#include <stdlib.h>
#include <time.h>
time_t time(time_t *t)
{
    const char *s = getenv("PN_FAKE_TIME");
    time_t v = (time_t)(s ? atoll(s) : 1);
    if (t) *t = v;
    return v;
}
