// background.cpp -- see background.h.
#include "background.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace {

// (mantissa, shared exponent) -> float, the conversion FreeImage's HDR plugin applies:
// value = mantissa * 2^(e - 136); e == 0 means black.
inline void rgbe_to_float(const unsigned char rgbe[4], float *rgb)
{
    if (rgbe[3] == 0) {
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
        return;
    }
    const float f = ldexpf(1.0f, (int)rgbe[3] - (128 + 8));
    rgb[0] = rgbe[0] * f;
    rgb[1] = rgbe[1] * f;
    rgb[2] = rgbe[2] * f;
}

bool read_scanline(FILE *fp, int width, std::vector<unsigned char> &line)
{
    line.resize((size_t)width * 4);
    unsigned char head[4];
    if (fread(head, 1, 4, fp) != 4)
        return false;
    const bool rle = width >= 8 && width < 32768 && head[0] == 2 && head[1] == 2 && !(head[2] & 0x80);
    if (!rle) {   // flat scanline (old-style repeat runs are not produced by any current writer)
        memcpy(line.data(), head, 4);
        return fread(line.data() + 4, 1, (size_t)(width - 1) * 4, fp) == (size_t)(width - 1) * 4;
    }
    if (((int)head[2] << 8 | head[3]) != width)
        return false;
    std::vector<unsigned char> plane((size_t)width);
    for (int channel = 0; channel < 4; channel++) {
        int x = 0;
        while (x < width) {
            int count = fgetc(fp);
            if (count == EOF)
                return false;
            if (count > 128) {   // a run of one value
                count -= 128;
                const int value = fgetc(fp);
                if (value == EOF || count == 0 || x + count > width)
                    return false;
                memset(plane.data() + x, value, (size_t)count);
            } else {             // literal bytes
                if (count == 0 || x + count > width || fread(plane.data() + x, 1, (size_t)count, fp) != (size_t)count)
                    return false;
            }
            x += count;
        }
        for (int i = 0; i < width; i++)
            line[(size_t)i * 4 + channel] = plane[i];
    }
    return true;
}

}   // namespace

bool read_radiance_hdr(const std::string &filename, float2Dimage &image)
{
    FILE *fp = fopen(filename.c_str(), "rb");
    if (!fp) {
        fprintf(stderr, "Failed to load image from %s\n", filename.c_str());
        return false;
    }
    char line[512];
    bool magic = false, format_ok = false;
    int width = 0, height = 0;
    // header: "#?RADIANCE" / "#?RGBE", key=value lines, a blank line, then the resolution
    while (fgets(line, sizeof(line), fp)) {
        if (line[0] == '\n' || (line[0] == '\r' && line[1] == '\n'))
            break;
        if (!strncmp(line, "#?", 2))
            magic = true;
        if (!strncmp(line, "FORMAT=32-bit_rle_rgbe", 22))
            format_ok = true;
    }
    if (!fgets(line, sizeof(line), fp) || sscanf(line, "-Y %d +X %d", &height, &width) != 2 || width <= 0 || height <= 0 ||
        !magic || !format_ok) {
        fprintf(stderr, "%s: not a top-down 32-bit_rle_rgbe Radiance picture\n", filename.c_str());
        fclose(fp);
        return false;
    }
    image.width = width;
    image.height = height;
    image.pixels.assign((size_t)width * height * 3, 0.0f);
    std::vector<unsigned char> scan;
    for (int j = 0; j < height; j++) {
        if (!read_scanline(fp, width, scan)) {
            fprintf(stderr, "%s: scanline %d is damaged\n", filename.c_str(), j);
            fclose(fp);
            return false;
        }
        // the file runs top to bottom; row 0 of the image is the bottom row
        float *dst = image.pixels.data() + (size_t)(height - 1 - j) * width * 3;
        for (int i = 0; i < width; i++)
            rgbe_to_float(&scan[(size_t)i * 4], dst + (size_t)i * 3);
    }
    fclose(fp);
    return true;
}

bool load_background(const std::string &spec, float2Dimage &image)
{
    float rf, gf, bf;
    unsigned int rx, gx, bx;
    if (sscanf(spec.c_str(), "%f, %f, %f", &rf, &gf, &bf) == 3) {
        image.width = image.height = 1;
        image.pixels = {rf, gf, bf};
        return true;
    }
    if (spec == "grid") {
        const int width = 2048, height = width / 2, tile = 8, bar = 1;
        image.width = width;
        image.height = height;
        image.pixels.assign((size_t)width * height * 3, 0.0f);
        for (int j = 0; j < height; j++)
            for (int i = 0; i < width; i++)
                if ((i % tile) < bar || (j % tile) < bar) {
                    float *px = image.pixels.data() + 3 * ((size_t)width * j + i);
                    px[0] = px[1] = px[2] = 1.0f;
                }
        return true;
    }
    if (spec.size() == 6 && sscanf(spec.c_str(), "%2x%2x%2x", &rx, &gx, &bx) == 3) {
        image.width = image.height = 1;
        image.pixels = {rx / 255.0f, gx / 255.0f, bx / 255.0f};
        return true;
    }
    return read_radiance_hdr(spec, image);
}
