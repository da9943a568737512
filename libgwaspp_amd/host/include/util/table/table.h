// util/table/table.h -- the storage contract every genotype table builds on (reference:
// src/util/table/table.h:49-74): a table is sized by two borrowed indexers -- one entry per row
// (SNP), one per column (sample), read through included_size() -- and owns a flat `data` array of
// D.  The HBM-resident table keeps `data` NULL: its cells live on the device.
#pragma once
#include <string>

#include "util/index_set/indexer.h"

namespace util {

template <class D>
class Table {
public:
    Table(indexer *row_index, indexer *column_index)
        : rows(row_index),
          columns(column_index),
          max_row(row_index->included_size()),
          max_column(column_index->included_size()),
          data(nullptr) {}
    virtual ~Table() { delete[] data; }

    // dimensions and labels
    int row_size() const { return max_row; }
    int column_size() const { return max_column; }
    const indexer *getRowSet() { return rows; }
    const indexer *getColumnSet() { return columns; }
    std::string rowID(int i) const { return rows->getIDAtOrderedIndex(i); }
    std::string columnID(int i) const { return columns->getIDAtOrderedIndex(i); }

    // cell access; packed tables override it
    virtual D operator()(int r, int c) { return data[static_cast<long>(r) * max_column + c]; }

protected:
    virtual void initialize() = 0;  // geometry + allocation, called from the subclass constructor

    indexer *rows, *columns;
    int max_row, max_column;
    D *data;
};

}  // namespace util
