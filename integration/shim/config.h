/* Build-time config for compiling the UNMODIFIED reference translation units
 * out of /root/reference/src (test infrastructure only -- see oracle/Makefile).
 * Stands in for the file CMake would generate from src/config.h.in:1-25. */
#pragma once
#define HAVE_STRNDUP
#define HAVE_CMPLXF
#define HAVE_IMAGINARY_I
#define HAVE_COMPLEX_I
#define LIBRARY_DEBUG_LEVEL 5
