/* Compatibility forwarder: the reference splits its API over include/huffman/histogram.h;
 * here every declaration lives in include/huffman.h. */
#include "../huffman.h"
