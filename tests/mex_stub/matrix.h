/* matrix.h -- stand-in (see mex.h in this directory): everything lives in mex.h. */
#include "mex.h"
