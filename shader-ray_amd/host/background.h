// background.h -- the environment image the tracer looks up.
//
// The reference parses its second command-line argument into a float RGB image
// (float2Dimage, ray.cpp:330-343; parsing at ray.cpp:1002-1075):
//     "r, g, b"   three floats          -> 1x1 image
//     "grid"      procedural 2048x1024, 8-pixel tiles with 1-pixel white bars
//     "rrggbb"    hex                   -> 1x1 image
//     otherwise   an image file, decoded by FreeImagePlus (FIT_RGBF or 8-bit)
// FreeImagePlus is not available here; the file branch reads Radiance RGBE (.hdr / .pic),
// which is what the reference's suggested environments (images/pisa.hdr, README.md:14) are.
// Row 0 of `pixels` is the BOTTOM row of the picture (FreeImage scanline order, and the
// order glTexImage2D consumes at ray.cpp:508), i.e. texture t = 0 = straight down.
#pragma once

#include <string>
#include <vector>

struct float2Dimage {
    int width = 0;
    int height = 0;
    std::vector<float> pixels;   // 3 floats per pixel, row-major from the bottom row
};

// Returns false (message on stderr) when the spec is none of the forms above or the file
// cannot be decoded.
bool load_background(const std::string &spec, float2Dimage &image);

// Radiance RGBE decoder: flat and run-length-encoded scanlines, "-Y h +X w" orientation.
bool read_radiance_hdr(const std::string &filename, float2Dimage &image);
