/* Compatibility forwarder: the reference splits its API over include/huffman/errors.h;
 * here every declaration lives in include/huffman.h. */
#include "../huffman.h"
