/* lislib.h -- umbrella header the reference's drivers include (reference include/lislib.h); here the public
 * API is all there is. */
#ifndef __LISLIB_H__
#define __LISLIB_H__
#include <string.h>
#include "lis.h"
#include "lis_amd.h"
#endif
