/* placeholder: replaced by the deflate restatement */
#include "spng_oracle.h"
