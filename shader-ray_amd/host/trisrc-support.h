// trisrc-support.h -- loader for the private "trisrc" text format
// (reference: trisrc-support.h:24, trisrc-support.cpp:43-105).
#pragma once

#include <cstdio>

#include "triangle-set.h"

// Appends every triangle of the stream to `triangles`.  Returns false (after
// a message on stderr) when a record is cut short; a stream that simply ends,
// or whose next record does not open with a quoted name, ends the parse
// successfully -- the reference's fscanf loop behaves the same way.
bool ParseTriSrc(FILE *fp, triangle_set_ptr triangles);

// Same parser over an in-memory, NUL-terminated text.
bool ParseTriSrcText(const char *text, triangle_set_ptr triangles);
