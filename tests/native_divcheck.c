/* CPU rehearsal of csrc/exact_div.h with fmaf(): the 5-operation sequence must equal
 * true division for operands in the admitted ranges.  Built and run by
 * tests/test_exact_division_cpu.py.  Prints the mismatch count. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static float from_bits(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static float random_in(uint64_t bits, int lo, int hi, unsigned shape)
{
    uint32_t frac = (uint32_t)bits & 0x7fffffu;
    switch (shape & 7u) {
    case 1: frac = 0x7fffffu; break;
    case 2: frac = 0u; break;
    case 3: frac = 1u << ((bits >> 40) % 23); break;
    case 4: frac = 0x7fffffu ^ (1u << ((bits >> 40) % 23)); break;
    default: break;
    }
    int e = lo + (int)((bits >> 24) % (uint64_t)(hi - lo + 1));
    uint32_t sign = (uint32_t)(bits >> 63) << 31;
    return from_bits(sign | ((uint32_t)(e + 127) << 23) | frac);
}

int main(int argc, char **argv)
{
    uint64_t pairs = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000, seed = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
    uint64_t bad = 0;
    for (uint64_t i = 0; i < pairs; i++) {
        uint64_t h0 = mix64(seed + 2 * i), h1 = mix64(seed + 2 * i + 1);
        volatile float b = random_in(h0, -40, 19, (unsigned)(h0 >> 48));
        volatile float a = random_in(h1, -93, 60, (unsigned)(h1 >> 48));
        if (((h1 >> 56) & 63u) == 0u) a = 0.0f;
        if (((h1 >> 56) & 63u) == 1u) a = b * random_in(mix64(h1), -3, 3, (unsigned)(h1 >> 51));
        float y = 1.0f / b;
        float q0 = a * y;
        float r0 = fmaf(-b, q0, a);
        float q1 = fmaf(r0, y, q0);
        float r1 = fmaf(-b, q1, a);
        float fast = fmaf(r1, y, q1);
        float exact = a / b;
        /* four-operation form with a split reciprocal */
        float yl = fmaf(-b, y, 1.0f) * y;
        float p1 = fmaf(a, yl, a * y);
        float fast4 = fmaf(fmaf(-b, p1, a), y, p1);
        if (memcmp(&fast4, &exact, 4) != 0 && !(fast4 == 0.0f && exact == 0.0f)) {
            if (bad < 5) fprintf(stderr, "mismatch4 a=%a b=%a fast4=%a exact=%a\n", a, b, fast4, exact);
            bad++;
        }
        if (memcmp(&fast, &exact, 4) != 0 && !(fast == 0.0f && exact == 0.0f)) {
            if (bad < 5) fprintf(stderr, "mismatch a=%a b=%a fast=%a exact=%a\n", a, b, fast, exact);
            bad++;
        }
    }
    printf("%llu\n", (unsigned long long)bad);
    return 0;
}
